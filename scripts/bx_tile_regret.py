"""The bf16-exact selector (dga_tiling_bf16_exact) against every tile / split-K of its menu on the reference's prefill-sized shapes and
a few with awkward round counts: regret of the pick."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

SHAPES = [(1024, 18432, 7168), (1024, 4096, 7168), (2048, 4096, 7168), (4096, 4096, 4096), (4096, 2048, 7168), (1024, 7168, 18432),
          (2048, 7168, 2048), (1536, 4096, 7168), (3072, 4096, 4096), (512, 7168, 4096), (768, 7168, 2048), (2304, 4096, 7168), (5120, 5120, 5120), (4096, 7168, 2048), (8064, 4096, 1024), (4096, 5120, 1536), (3584, 4096, 2048), (2048, 7168, 1024), (1536, 7168, 2048)]
TILES = [(128, 256), (64, 256), (128, 128), (64, 128)]
for (m, n, k) in SHAPES:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=3)
    o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    pick = dga.tiling(m, n, k, policy="bf16_exact")
    f0 = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=pick)
    f0(); torch.cuda.synchronize()
    ref = o.clone()
    us0 = min(bench._prewarmed_us(f0, 30, 100.0) for _ in range(2))
    rows = []
    for (bm, bn) in TILES:
        for sk in (1, 2, 4, 0):            # 0: the 128 x 256 tile with its last partial round in quarter tiles (kernelSerial 5)
            if sk == 0 and (bm, bn) != (128, 256):
                continue
            t = dga.tiling(m, n, k, policy="bf16_exact")
            t.m1, t.n1, t.splitkFactor = bm, bn, max(sk, 1)
            t.kernelSerial = 4 if sk > 1 else (5 if sk == 0 else 0)
            t.blockDim = ((m + bm - 1) // bm) * ((n + bn - 1) // bn) * max(sk, 1)
            t.stages = 3
            if dga.tiling_check(t) != 0:
                continue
            fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=t)
            try:
                fn(); torch.cuda.synchronize()
            except Exception as e:
                continue
            us = min(bench._prewarmed_us(fn, 30, 100.0) for _ in range(2))
            rows.append((us, bm, bn, sk))
    rows.sort()
    best = rows[0]
    print(f"{m:>5}x{n:>6}x{k:>6} pick {pick.m1}x{pick.n1} ks{pick.kernelSerial} s{pick.splitkFactor} st{pick.stages}: {us0:8.2f} us | best {best[1]}x{best[2]} s{best[3]} {best[0]:8.2f} us"
          f" | regret {100.0 * (us0 / best[0] - 1.0):5.1f} % | " + "  ".join(f"{r[1]}x{r[2]}s{r[3]}:{r[0]:.1f}" for r in rows[:5]), flush=True)
    del a, b, o, ref
