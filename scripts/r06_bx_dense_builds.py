"""The bf16-exact 128 x 256 tile on dense rasters: the persistent build (tiling.build 7), the one-tile build (8) and the masked-grouped
layout's kernel (9: self-contained k blocks, two of them in flight) -- same bits, one process, interleaved.
Usage: python scripts/r06_bx_dense_builds.py [M N K ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import deepgemm_ascend_amd as dga

shapes = [(4096, 4096, 4096), (4096, 2048, 7168), (1024, 4096, 7168), (8192, 8192, 8192)]
if len(sys.argv) > 3:
    v = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
for m, n, k in shapes:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    outs, res = {}, {}
    fns = {}
    for build in (7, 8, 9):
        t = dga.tiling(m, n, k, policy="bf16_exact")
        t.m1, t.n1, t.kernelSerial, t.splitkFactor, t.stages, t.build = 128, 256, 0, 1, 3, build
        o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        fns[build] = (lambda t=t, o=o: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, policy="bf16_exact", tiling_=t))
        fns[build](); torch.cuda.synchronize()
        outs[build] = o
    same = all(torch.equal(outs[7].view(torch.int16), outs[x].view(torch.int16)) for x in (8, 9))
    for rnd in range(3):
        for build in (7, 8, 9):
            us = bench._prewarmed_us(fns[build], 50, 80.0)
            res[build] = min(res.get(build, 1e9), us)
    print(f"{m}x{n}x{k}: persistent {res[7]:8.2f} us  one-tile {res[8]:8.2f} us  grouped-kernel {res[9]:8.2f} us  same bits: {same}", flush=True)
