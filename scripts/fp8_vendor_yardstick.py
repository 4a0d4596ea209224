"""A yardstick for the dense fp8 rate on this part: the vendor library's fp8 GEMM as torch._scaled_mm reaches it (hipBLASLt; ONE fp32 scale
per tensor -- no 1x128 / 128x128 block scales, no per-k-block promotion, so less work per flop than gemm_fp8_fp8_bf16_nt does) beside the
operator's fast policy on the same shapes; warm, device time by graph replay.  Nothing in the product calls a library.
Usage: python scripts/fp8_vendor_yardstick.py"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402


def main():
    for (m, n, k) in [(4096, 4096, 4096), (4096, 2048, 7168), (8192, 8192, 8192), (2048, 4096, 7168), (1024, 4096, 7168), (1024, 18432, 7168)]:
        a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        ours = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out)
        a8 = a.view(torch.float8_e4m3fn) if a.dtype != torch.float8_e4m3fn else a
        b8 = b.view(torch.float8_e4m3fn) if b.dtype != torch.float8_e4m3fn else b
        one = torch.ones((), device="cuda", dtype=torch.float32)
        row = {"shape": [m, n, k]}
        try:
            vend = lambda: torch._scaled_mm(a8, b8.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
            vend(); torch.cuda.synchronize()
            t = {}
            for rnd in range(3):
                for name, fn in (("ours_us", ours), ("vendor_tensorwise_us", vend)):
                    t[name] = min(t.get(name, 1e30), sweep.graph_us(fn, 10, replays=3, prewarm_ms=100.0))
            row.update({k2: round(v, 1) for k2, v in t.items()})
            row["ours_tflops"] = round(2.0 * m * n * k / t["ours_us"] / 1e6)
            row["vendor_tflops"] = round(2.0 * m * n * k / t["vendor_tensorwise_us"] / 1e6)
        except Exception as e:   # noqa: BLE001
            row["vendor_error"] = repr(e)[:200]
            row["ours_us"] = round(min(sweep.graph_us(ours, 10, replays=3, prewarm_ms=100.0) for _ in range(2)), 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
