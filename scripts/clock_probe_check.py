"""Is the loop-clock build representative of the product kernel?  Launch interval of both, back to back (development aid)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
from widen_perf import timeit
for (m, n, k) in [(4096, 4096, 4096), (4096, 2048, 7168)]:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    t = dga.tiling(m, n, k)
    fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
    for rnd in range(3):
        us = timeit(fn, iters=400, warm=400)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mhz, loop_us = dga.gemm_fp8_loop_clock((a, sfa), (b, sfb), out, tiling_=t, launches=1000)
        wall = (time.perf_counter() - t0) * 1e6 / 1000
        print(f"{m}x{n}x{k}: product {us:.1f} us/launch | clock build {wall:.1f} us/launch (wall, 1000 launches), loop {loop_us:.1f} us at {mhz:.0f} MHz", flush=True)
