"""Shapes outside the predictor's training range (N or K < 512, tiny everything): default tiling vs the heuristic vs the
best of the swept candidates (development aid)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import sweep
shapes = [(8192, 128, 7168), (128, 128, 16384), (4096, 256, 256), (256, 256, 256), (16, 16, 128), (7, 24, 48), (16384, 256, 128),
          (64, 128, 128), (2048, 384, 4096), (333, 200, 1000), (8192, 8192, 128), (1, 7168, 7168), (4, 256, 16384)]
for (m, n, k) in shapes:
    a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    def run(t):
        fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
        fn(); torch.cuda.synchronize()
        ok, _ = sweep.is_correct(golden, out, s_abs, short_k=k < 128)
        return min(sweep.time_us(fn, warm=5, iters=30) for _ in range(3)), ok
    td, okd = run(dga.tiling(m, n, k))
    th, okh = run(dga.select_kernel(m, n, k))
    best = (1e30, None)
    if k % 16 == 0:
        for p in sweep.candidates(m, n, k, [0]):
            t = dga.select_kernel(m, n, k)
            t.m1, t.n1, t.swizzleOffset, t.stages, t.wavesM, t.wavesN, t.dispatchPolicyTag = p["m1"], p["n1"], p["raster"], p["stages"], 0, 0, p["policy"]
            t.splitkFactor = p["splitk"]; t.kernelSerial = 5 if p.get("tail") else (4 if p["splitk"] > 1 else 0)
            us, ok = run(t)
            if ok and us < best[0]: best = (us, p)
    d = dga.tiling(m, n, k)
    print(f"{m}x{n}x{k}: default {td:.1f} us ({d.m1}x{d.n1} st{d.stages} sk{d.splitkFactor} ok={okd})  heuristic {th:.1f}  best {best[0]:.1f} {best[1]}", flush=True)
