"""configs[3] under the in-contract policy: every build of the bf16-exact arithmetic that takes the masked grouped layout
(tiling.stages names the build: 3 in-register, 7 persistent, 8 one-tile, 4 A-image, 5 / 6 both-operand images on 8 / 4 waves)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

g, mmax, n, k = 256, 128, 2048, 7168
gen = torch.Generator(device="cuda").manual_seed(5)
xb = torch.randn((g, n, k), device="cuda", generator=gen)
sb = xb.view(g, n // 128, 128, k // 128, 128).abs().amax(dim=(2, 4)).clamp_min(1e-30) / 448.0
qb = (xb.view(g, n // 128, 128, k // 128, 128) / sb[:, :, None, :, None]).reshape(g, n, k).to(torch.float8_e4m3fn).view(torch.uint8)
del xb
xa = torch.randn((g, mmax, k), device="cuda", generator=gen)
sa = xa.view(g, mmax, k // 128, 128).abs().amax(dim=3).clamp_min(1e-30) / 448.0
qa = (xa.view(g, mmax, k // 128, 128) / sa[..., None]).reshape(g, mmax, k).to(torch.float8_e4m3fn).view(torch.uint8)
del xa
ref = None
for name, mask in (("full", torch.full((g,), mmax, dtype=torch.int32, device="cuda")),
                   ("random", torch.randint(0, mmax + 1, (g,), generator=torch.Generator().manual_seed(99)).to(torch.int32).cuda())):
    ref = None
    for stages in (3, 7, 8, 4, 5, 6):
        t = dga.tiling(mmax, n, k, groups=g, expected_m=mmax, policy="bf16_exact")
        t.stages = stages
        o = torch.zeros((g, mmax, n), dtype=torch.bfloat16, device="cuda")
        fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((qa, sa), (qb, sb), o, mask, mmax, tiling_=t)
        try:
            fn(); torch.cuda.synchronize()
        except Exception as e:
            print(f"mask {name} stages {stages}: {e}")
            continue
        same = None if ref is None else bool(torch.equal(ref.view(torch.int16), o.view(torch.int16)))
        if ref is None:
            ref = o.clone()
        us = min(bench._prewarmed_us(fn, 20, 100.0) for _ in range(2))
        print(f"mask {name} tile {t.m1}x{t.n1} stages {stages}: {us:8.1f} us  same bytes as stages 3: {same}", flush=True)
