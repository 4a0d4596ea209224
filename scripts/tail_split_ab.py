"""Tail split (kernelSerial 5) vs one launch over all tiles, same device, same inputs (development aid)."""
import os, sys
os.environ["DGA_NO_TUNED_TABLE"] = "1"
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
from widen_perf import timeit

for (m, n, k) in [(1024, 18432, 7168), (4352, 4096, 4096), (2304, 8192, 4096), (4096, 4608, 7168), (5000, 4096, 2048)]:
    a, sfa, b, sfb = bench.make_dense_inputs(m // 128 * 128, n, k, seed=0) if m % 128 == 0 else bench.make_dense_inputs(5120, n, k, seed=0)
    if m % 128: a, sfa = a[:m].contiguous(), sfa[:m].contiguous()
    outs = {}
    t = dga.tiling(m, n, k)
    for name in ("tail-split", "plain"):
        tt = dga.tiling(m, n, k)
        if name == "plain":
            tt.kernelSerial = 0; tt.splitkFactor = 1
        out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
        fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=tt)
        us = min(timeit(fn, iters=20, warm=5) for _ in range(3))
        outs[name] = out
        print(f"{m}x{n}x{k} {name}: serial {tt.kernelSerial} splitk {tt.splitkFactor} tile {tt.m1}x{tt.n1}: {us:.1f} us  {2.0*m*n*k/us/1e6:.0f} TFLOP/s", flush=True)
    d = (outs["tail-split"].float() - outs["plain"].float()).abs().max().item()
    print("   max abs diff between the two:", d, " equal bytes:", torch.equal(outs["tail-split"].view(torch.int16), outs["plain"].view(torch.int16)), flush=True)
