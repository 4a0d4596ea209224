import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
from widen_perf import timeit
for (m, n, k) in [(4096, 4096, 4096), (4096, 2048, 7168), (128, 4096, 7168)]:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, strict=True)
    us = min(timeit(fn, iters=5, warm=2) for _ in range(2))
    print(f"strict {m}x{n}x{k}: {us:.0f} us  {2.0 * m * n * k / us / 1e6:.1f} TFLOP/s", flush=True)
