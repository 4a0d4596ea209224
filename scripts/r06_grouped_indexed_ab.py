"""The indexed masked-grouped GEMM (rows found through the slot table) under the bf16-exact policy: the layout's own kernel (build 9)
against the one-tile build (8) the indexed form ran until round 6, and the packed call beside them; configs[3], one process."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import deepgemm_ascend_amd as dga

G, MMAX, N, K = 256, 128, 2048, 7168
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randint(0, 120, (G * MMAX, K), dtype=torch.uint8, device="cuda", generator=g)
b = torch.randint(0, 120, (G, N, K), dtype=torch.uint8, device="cuda", generator=g)
sfa = torch.rand((G * MMAX, K // 128), device="cuda") + 0.5
sfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
out = torch.zeros((G * MMAX, N), dtype=torch.bfloat16, device="cuda")
perm = torch.randperm(G * MMAX, device="cuda", generator=g)          # the rows of an expert lie anywhere in the flat buffer
row_index = perm.to(torch.int64).contiguous()
cpu = torch.Generator().manual_seed(99)
for name, masked in (("full", torch.full((G,), MMAX, dtype=torch.int32)), ("random_0_128", torch.randint(0, MMAX + 1, (G,), generator=cpu).to(torch.int32)),
                     ("random_0_16", torch.randint(0, 17, (G,), generator=cpu).to(torch.int32))):
    masked = masked.cuda()
    res = {}
    for build in (9, 8):
        t = dga.tiling(MMAX, N, K, groups=G, expected_m=MMAX, policy="bf16_exact")
        t.build = build
        fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(a, sfa, 0, K // 128, (b, sfb), out, row_index, masked, MMAX, MMAX,
                                                                         policy="bf16_exact", tiling_=t)
        fn(); torch.cuda.synchronize()
        res[f"indexed build {build}"] = min(bench._prewarmed_us(fn, 30, 100.0) for _ in range(2))
    t = dga.tiling(MMAX, N, K, groups=G, expected_m=MMAX, policy="bf16_exact")
    fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a.view(G, MMAX, K), sfa.view(G, MMAX, -1)), (b, sfb), out.view(G, MMAX, N), masked, MMAX,
                                                           policy="bf16_exact", tiling_=t)
    fn(); torch.cuda.synchronize()
    res["packed build 9"] = min(bench._prewarmed_us(fn, 30, 100.0) for _ in range(2))
    print(name, {k: round(v, 1) for k, v in res.items()}, flush=True)
