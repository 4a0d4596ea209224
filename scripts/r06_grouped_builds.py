import sys
sys.path.insert(0, "/root/repo")
import torch, bench
import deepgemm_ascend_amd as dga
G, MMAX, N, K = 256, 128, 2048, 7168
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randint(0, 120, (G, MMAX, K), dtype=torch.uint8, device="cuda", generator=g)
b = torch.randint(0, 120, (G, N, K), dtype=torch.uint8, device="cuda", generator=g)
sfa = torch.rand((G, MMAX, K // 128), device="cuda") + 0.5
sfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
out = torch.zeros((G, MMAX, N), dtype=torch.bfloat16, device="cuda")
cpu = torch.Generator().manual_seed(99)
for name, masked in (("full", torch.full((G,), MMAX, dtype=torch.int32)), ("random", torch.randint(0, MMAX + 1, (G,), generator=cpu).to(torch.int32)),
                     ("r65_128", torch.randint(65, MMAX + 1, (G,), generator=cpu).to(torch.int32))):
    masked = masked.cuda()
    res = {}
    for rnd in range(2):
        for build in (9, 4, 7, 5):
            t = dga.tiling(MMAX, N, K, groups=G, expected_m=MMAX, policy="bf16_exact")
            t.build = build
            fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, MMAX, policy="bf16_exact", tiling_=t)
            fn(); torch.cuda.synchronize()
            res.setdefault(build, []).append(round(bench._prewarmed_us(fn, 30, 100.0), 1))
    print(name, res, flush=True)
