"""Dense schedules / tiles A/B at sustained clocks: every variant gets its own 300 ms warm run, then 3 x 200 timed
launches (development aid)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

def steady(fn, ms=300, steps=200):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(50): fn()
        torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / steps)
    return best

shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 2048, 7168)]
if len(sys.argv) > 3: shapes = [tuple(int(v) for v in sys.argv[1:4])]
for (m, n, k) in shapes:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    for name, (bm, bn, wm, wn, st, pol) in {"256x256 plain": (256, 256, 4, 2, 2, 0), "256x256 ping-pong": (256, 256, 4, 2, 2, 1),
                                             "256x256 continuous": (256, 256, 4, 2, 2, 2), "128x256 st3": (128, 256, 2, 2, 3, 0),
                                             "128x256 st2": (128, 256, 2, 2, 2, 0), "128x256 8 waves": (128, 256, 2, 4, 2, 0), "128x256 8 waves st3": (128, 256, 2, 4, 3, 0), "128x256 cont": (128, 256, 2, 2, 2, 2), "128x256 8 waves cont": (128, 256, 2, 4, 2, 2),
                                             "256x128": (256, 128, 4, 1, 2, 0), "128x128 st3": (128, 128, 2, 2, 3, 0)}.items():
        t = dga.tiling(m, n, k)
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = bm, bn, wm, wn, st, pol, 0, 1
        us = steady(lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t), steps=200 if m * n * k < 2 ** 37 else 40)
        print(f"{m}x{n}x{k} {name}: {us:.1f} us  {2.0*m*n*k/us/1e6:.0f} TFLOP/s", flush=True)
