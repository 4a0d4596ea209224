"""A/B of the output-store cache policy on dense rasters ($DGA_OUT_NT = 0 / 1 / 2 / 3 = plain / nt / sc0 sc1 / sc0 sc1 nt), one child
process per setting (the switch is read once per process): launch interval of back-to-back calls, fast and bf16-exact policies."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, str(ROOT))
    import torch
    import deepgemm_ascend_amd as dga
    import bench
    for (m, n, k) in [(4096, 4096, 4096), (4096, 2048, 7168), (8192, 8192, 4096)]:
        a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=1, ue8m0=True)
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        row = []
        for pol in ("fast", "fast_ue8m0", "bf16_exact"):
            fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy=pol)
            fn(); torch.cuda.synchronize()
            us = min(bench._prewarmed_us(fn, 300, 300.0) for _ in range(2))
            row.append(f"{pol} {us:7.2f} us")
        print(f"DGA_OUT_NT={os.environ.get('DGA_OUT_NT', '-')}  {m}x{n}x{k}: " + "   ".join(row), flush=True)
        del a, b, out
else:
    for v in ("0", "1", "2", "3", "0", "1"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DGA_OUT_NT=v), check=True)
