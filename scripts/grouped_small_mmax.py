"""Grouped masked-M GEMM with a small m_max (capacity per expert): tile builds side by side (development aid)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd import parallel
from widen_perf import timeit
g = torch.Generator(device="cuda").manual_seed(0)
G, N, K = 256, 2048, 7168
b = parallel._rand_fp8((G, N, K), g, "cuda"); sfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
for MM in (16, 32, 64):
    a = parallel._rand_fp8((G, MM, K), g, "cuda"); sfa = torch.rand((G, MM, K // 128), device="cuda") + 0.5
    out = torch.empty((G, MM, N), dtype=torch.bfloat16, device="cuda")
    mask = torch.full((G,), MM, dtype=torch.int32, device="cuda")
    byt = G * N * K + G * MM * (K + 224 + 2 * N)
    for name, (bm, bn, wm, wn, st) in {"auto": (0, 0, 0, 0, 0), "128x256 st3": (128, 256, 2, 2, 3), "64x256 st3": (64, 256, 1, 4, 3), "32x256": (32, 256, 1, 4, 2), "16x256": (16, 256, 1, 4, 2)}.items():
        t = dga.select_kernel(MM, N, K, groups=G, expected_m=MM)
        if bm: t.m1, t.n1, t.wavesM, t.wavesN, t.stages = bm, bn, wm, wn, st
        us = min(timeit(lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, mask, MM, tiling_=t), iters=10, warm=3) for _ in range(3))
        print(f"m_max {MM} {name}: tile {t.m1}x{t.n1} st{t.stages}: {us:.0f} us  {byt/us/1e3:.0f} GB/s", flush=True)
