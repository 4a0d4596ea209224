"""The bf16-exact 128 x 256 tile on dense rasters that give every CU the same number of tiles: the persistent kernel (tiling.stages = 7)
against the one-tile build (8) -- the dispatcher's rule for when the persistent form pays (dga_launch.hip)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

for (m, n, k) in [(4096, 4096, 4096), (8192, 8192, 8192), (1024, 16384, 4096), (4096, 8192, 4096), (2048, 8192, 7168), (4096, 4096, 7168), (4096, 4096, 2048), (1024, 24576, 1536), (8192, 4096, 512)]:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=3)
    o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    res = {}
    for rep in range(2):
        for st in (7, 8):
            t = dga.tiling(m, n, k, policy="bf16_exact"); t.m1, t.n1, t.kernelSerial, t.splitkFactor, t.stages = 128, 256, 0, 1, st
            fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=t)
            fn(); torch.cuda.synchronize()
            us = min(bench._prewarmed_us(fn, 50, 200.0) for _ in range(2))
            res[st] = min(res.get(st, 1e9), us)
    tiles = (m // 128) * (n // 256)
    print(f"{m}x{n}x{k} tiles {tiles} ({tiles / 256:.2f} rounds): persistent {res[7]:8.2f} us  one-tile {res[8]:8.2f} us  ratio {res[8] / res[7]:.3f}", flush=True)
    del a, b, o
