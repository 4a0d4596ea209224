"""Time the auto-tiled fp8 GEMM on the reference's sweep shape list (benchmark.py:24-44)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import sweep
for (m, n, k) in sweep.SHAPE_GROUP:
    a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    t = dga.tiling(m, n, k)
    fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
    fn(); torch.cuda.synchronize()
    ok, diff = sweep.is_correct(golden, out, s_abs, short_k=k < 128)
    iters = 3 if k % 16 else 20
    us = sweep.time_us(fn, warm=2, iters=iters)
    byt = m * k + n * k + 2 * m * n
    print(f"{m:5d} x {n:5d} x {k:5d}  tile {t.m1:3d}x{t.n1:3d} st{t.stages} blocks {t.blockDim:5d}  {us:9.1f} us  {2.0*m*n*k/us/1e6:7.1f} TF  {byt/us/1e3:7.1f} GB/s  ok={ok}", flush=True)
