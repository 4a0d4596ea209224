import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
m, n, k = 4096, 4096, 4096
a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
for gm in [1, 2, 4, 8, 16]:
    t = dga.tiling(m, n, k); t.swizzleOffset = gm
    for _ in range(10): dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    print(f"xcd_remap={os.environ.get('DGA_XCD_REMAP','1')} raster_group={gm}: {us:.1f} us {2*m*n*k/us/1e6:.0f} TF", flush=True)
