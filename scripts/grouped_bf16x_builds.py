import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import deepgemm_ascend_amd as dga
from scripts.policy_perf import time_us
G, MMAX, N, K = 256, 128, 2048, 7168
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randint(0, 120, (G, MMAX, K), dtype=torch.uint8, device="cuda", generator=g)
b = torch.randint(0, 120, (G, N, K), dtype=torch.uint8, device="cuda", generator=g)
sfa = torch.rand((G, MMAX, K // 128), device="cuda") + 0.5
sfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
outs = []
for mask, masked in (("full", torch.full((G,), MMAX, dtype=torch.int32, device="cuda")),
                     ("random", torch.randint(0, 129, (G,), dtype=torch.int32, device="cuda", generator=g))):
    rows = int(masked.sum()); byt = G * N * K + rows * (K + 4 * (K // 128) + 2 * N)
    ref = None
    for name, st, wv in (("in-register", 3, (2, 4)), ("a-image", 4, (2, 4)), ("image8", 5, (2, 4)), ("image4", 6, (2, 4))):
        t = dga.tiling(MMAX, N, K, groups=G, expected_m=MMAX, policy="bf16_exact")
        t.m1, t.n1, t.stages, t.wavesM, t.wavesN = 128, 256, st, wv[0], wv[1]
        out = torch.zeros((G, MMAX, N), dtype=torch.bfloat16, device="cuda")
        fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, MMAX, tiling_=t, policy="bf16_exact")
        us = time_us(fn, 20, 100)
        same = None if ref is None else bool(torch.equal(out.view(torch.int16), ref.view(torch.int16)))
        if ref is None: ref = out.clone()
        print(mask, name, round(us, 1), "us", round(byt / us / 8e6, 3), "of 8 TB/s", "same bytes" if same else same, flush=True)
