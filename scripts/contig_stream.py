"""Contiguous-grouped layout in its HBM-bound regime (one 128-row block per expert): 4-wave vs 8-wave 3-stage builds."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd import parallel
from widen_perf import timeit
g = torch.Generator(device="cuda").manual_seed(0)
G, per, N, K = 256, 128, 2048, 7168
a = parallel._rand_fp8((G * per, K), g, "cuda"); b = parallel._rand_fp8((G, N, K), g, "cuda")
sfa = torch.rand((G * per, K // 128), device="cuda") + 0.5; sfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
idx = torch.arange(G, device="cuda", dtype=torch.int32).repeat_interleave(per).contiguous()
out = torch.empty((G * per, N), dtype=torch.bfloat16, device="cuda")
byt = G * N * K + G * per * (K + 224 + 2 * N)
for name, (wm, wn, ras) in {"auto": (0, 0, 0), "4 waves": (2, 2, 0), "8 waves": (2, 4, 0), "4 waves raster 1": (2, 2, 1), "8 waves raster 1": (2, 4, 1), "8 waves raster 2": (2, 4, 2)}.items():
    t = dga.tiling(G * per, N, K, groups=G, contiguous=True)
    if wm: t.wavesM, t.wavesN = wm, wn
    if ras: t.swizzleOffset = ras
    us = min(timeit(lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), out, idx, tiling_=t), iters=10, warm=3) for _ in range(3))
    print(f"{name}: raster {t.swizzleOffset} tile {t.m1}x{t.n1} waves {t.wavesM}x{t.wavesN} st{t.stages}: {us:.0f} us  {byt/us/1e3:.0f} GB/s  {2.0*G*per*N*K/us/1e6:.0f} TFLOP/s", flush=True)
