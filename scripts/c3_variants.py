"""BASELINE configs[2] (4096 x 2048 x 7168): tile builds side by side (development aid)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
from widen_perf import timeit
m, n, k = 4096, 2048, 7168
a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
for name, (bm, bn, wm, wn, st, pol) in {"auto": (0,) * 6, "128x256 (2,2) st3": (128, 256, 2, 2, 3, 0), "128x256 (2,2) st2": (128, 256, 2, 2, 2, 0),
                                         "128x256 (2,4) 8 waves": (128, 256, 2, 4, 2, 0), "256x128 (4,1)": (256, 128, 4, 1, 2, 0),
                                         "256x256 cont": (256, 256, 4, 2, 2, 2), "128x128 st3": (128, 128, 2, 2, 3, 0),
                                         "128x256 (2,4) st3 8 waves": (128, 256, 2, 4, 3, 0), "128x256 (2,4) cont": (128, 256, 2, 4, 2, 2),
                                         "128x256 (2,2) cont": (128, 256, 2, 2, 2, 2), "256x128 (4,1) cont": (256, 128, 4, 1, 2, 2),
                                         "128x128 cont": (128, 128, 2, 2, 2, 2),
                                         "128x256 (2,2) st3 + loader waves": (128, 256, 2, 2, 3, 4)}.items():
    t = dga.tiling(m, n, k)
    if bm:
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = bm, bn, wm, wn, st, pol, 0, 1
    fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
    us = min(timeit(fn, iters=100, warm=200) for _ in range(3))
    print(f"{name}: tile {t.m1}x{t.n1} waves {t.wavesM}x{t.wavesN} stages {t.stages}: {us:.1f} us  {2.0*m*n*k/us/1e6:.0f} TFLOP/s", flush=True)
