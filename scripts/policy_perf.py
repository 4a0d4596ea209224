"""Times the three arithmetic policies (fast / bf16_exact / strict) on BASELINE configs[1] and [2] in one process at
sustained clocks, and prints each one's distance from the strict kernel over all outputs (bench.parity_vs_strict).
Usage: python scripts/policy_perf.py [--tiles] (--tiles: every build of the bf16-exact menu on configs[1])"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import deepgemm_ascend_amd as dga  # noqa: E402


def time_us(fn, iters=200, prewarm_ms=400):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < prewarm_ms:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    res = {}
    for name in ("dense_4096", "dsv3_prefill"):
        m, n, k = bench.WORKLOADS[name]
        a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        row = {}
        for pol, iters in (("fast", 400), ("bf16_exact", 200), ("strict", 20)):
            fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy=pol)
            us = time_us(fn, iters, 400 if pol != "strict" else 100)
            fn(); torch.cuda.synchronize()
            par = bench.parity_vs_strict(dga, a, sfa, b, sfb, out)
            row[pol] = {"us": round(us, 2), "tflops": round(2.0 * m * n * k / us / 1e6, 1), "max_ulp": par["max_ulp"],
                        "frac_gt_2ulp": par["frac_gt_2ulp"], "worst_excess_over_S": par["worst_excess_over_S"]}
            print(name, pol, row[pol], flush=True)
        if "--tiles" in sys.argv and name == "dense_4096":
            for m1, n1 in ((128, 256), (128, 128), (64, 256), (64, 128), (32, 128)):
                t = dga.tiling(m, n, k)
                t.m1, t.n1, t.splitkFactor, t.kernelSerial = m1, n1, 1, 0
                for rg in (1, 2, 4, 8):
                    t.swizzleOffset = rg
                    fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact", tiling_=t)
                    us = time_us(fn, 100, 150)
                    print(f"  bf16_exact tile {m1}x{n1} raster {rg}: {us:.1f} us  {2.0 * m * n * k / us / 1e6:.0f} TFLOP/s", flush=True)
        res[name] = row
    print(json.dumps(res))


if __name__ == "__main__":
    main()
