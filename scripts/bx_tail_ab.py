"""The bf16-exact 128 x 256 tile with its last partial round in quarter tiles (kernelSerial 5: a second launch of 64 x 128 tiles) against
the single launch: time and bytes, dense shapes whose tile count leaves a tail of at most half the CUs."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

SHAPES = [(1024, 18432, 7168), (3072, 4096, 4096), (2304, 4096, 7168), (4608, 4096, 7168), (1024, 16384, 4096), (2560, 4096, 4096),
          (4096, 4096, 4096), (1152, 8192, 7168), (1280, 7168, 4096), (5120, 5120, 5120), (1024, 24576, 1536), (1279, 5120, 7680)]
for (m, n, k) in SHAPES:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=3)
    o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    pick = dga.tiling(m, n, k, policy="bf16_exact")
    f0 = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=pick)
    f0(); torch.cuda.synchronize()
    ref = o.clone()
    us0 = min(bench._prewarmed_us(f0, 30, 100.0) for _ in range(2))
    t = dga.tiling(m, n, k, policy="bf16_exact")
    t.m1, t.n1, t.splitkFactor, t.kernelSerial, t.stages = 128, 256, 1, 5, 3
    tiles = ((m + 127) // 128) * ((n + 255) // 256)
    tail = tiles % 256
    t.blockDim = tiles - tail + 4 * tail
    if dga.tiling_check(t) != 0:
        print(f"{m}x{n}x{k}: tail tiling refused"); continue
    o.zero_()
    fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=t)
    fn(); torch.cuda.synchronize()
    same = bool(torch.equal(o.view(torch.int16), ref.view(torch.int16))) if (pick.m1, pick.n1, pick.splitkFactor) == (128, 256, 1) else None
    us = min(bench._prewarmed_us(fn, 30, 100.0) for _ in range(2))
    print(f"{m:>5}x{n:>6}x{k:>6} tiles {tiles:>5} rounds {tiles / 256:5.2f} tail {tail:>3} | pick {pick.m1}x{pick.n1} ks{pick.kernelSerial} s{pick.splitkFactor}: {us0:8.2f} us | "
          f"128x256 + quarter-tile tail: {us:8.2f} us  ratio {us / us0:.3f}  same bytes: {same}", flush=True)
    del a, b, o, ref
