"""Time AND fabric traffic of one dense problem under each raster group (tiling.swizzleOffset = tile-rows walked together; the XCD
remap then gives every XCD a patch of that height).  Parent: times every variant, then runs itself twice under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (separate passes, MI355X_MICROARCH.md) and reads the per-dispatch
counters back in launch order.  FETCH_SIZE x 2 (gfx950: 64 B counted per 128-B request) x 1024 + WRITE_SIZE x 1024 = bytes.
  python scripts/raster_traffic.py M N K policy [rasters...]"""
import csv
import os
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LAUNCHES = 24


def variants(argv):
    m, n, k, pol = int(argv[0]), int(argv[1]), int(argv[2]), argv[3]
    rasters = [int(x) for x in argv[4:]] or [1, 2, 4, 8, 16, 32]
    return m, n, k, pol, rasters


def child(argv, timed):
    sys.path.insert(0, str(ROOT))
    import torch
    import deepgemm_ascend_amd as dga
    import bench
    m, n, k, pol, rasters = variants(argv)
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    base = dga.tiling(m, n, k, policy=pol if pol == "bf16_exact" else None)
    for r in rasters:
        t = dga.tiling(m, n, k, policy=pol if pol == "bf16_exact" else None)
        t.swizzleOffset = r
        fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
        if timed:
            us = min(bench._prewarmed_us(fn, 200, 200.0) for _ in range(2))
            print(f"raster {r:3d}: {us:8.2f} us  (tile {t.m1}x{t.n1}, selector's raster {base.swizzleOffset})", flush=True)
        else:
            for _ in range(LAUNCHES):
                fn()
            torch.cuda.synchronize()


def pmc(argv, counter):
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    d = tempfile.mkdtemp(prefix="dga_raster_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, __file__, "pmc-child"] + argv
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        f = sorted(Path(d).rglob("*_counter_collection.csv"))[0]
        per = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "gemm_fp8" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    per[int(row["Dispatch_Id"])] = per.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
        return [v for _, v in sorted(per.items())]
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    if sys.argv[1] == "timed-child":
        child(sys.argv[2:], True)
    elif sys.argv[1] == "pmc-child":
        child(sys.argv[2:], False)
    else:
        argv = sys.argv[1:]
        m, n, k, pol, rasters = variants(argv)
        print(f"== {m} x {n} x {k}, policy {pol}: algorithmic bytes {m * k + n * k + 2 * m * n + 4 * (m * (k // 128) + (n // 128) * (k // 128))}", flush=True)
        subprocess.run([sys.executable, __file__, "timed-child"] + argv, check=True)
        fetch, write = pmc(argv, "FETCH_SIZE"), pmc(argv, "WRITE_SIZE")
        for i, r in enumerate(rasters):
            fs = fetch[i * LAUNCHES + 8:(i + 1) * LAUNCHES]
            ws = write[i * LAUNCHES + 8:(i + 1) * LAUNCHES]
            if fs and ws:
                fb, wb = sum(fs) / len(fs) * 2 * 1024, sum(ws) / len(ws) * 1024
                print(f"raster {r:3d}: fetch {fb / 1e6:8.1f} MB  write {wb / 1e6:7.1f} MB  total {(fb + wb) / 1e6:8.1f} MB", flush=True)
