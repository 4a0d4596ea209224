"""The 16-bit operator (catlass_dynamic_matmul, bf16, NT) beside the vendor GEMM library as torch.matmul reaches it (hipBLASLt /
Tensile), on the reference's 18-shape list: device time by graph replay, warm.  A yardstick only -- nothing in the product calls a
GEMM library.  Usage: python scripts/op16_vs_vendor.py"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402


def main():
    for (m, n, k) in sweep.SHAPE_GROUP:
        g = torch.Generator(device="cuda").manual_seed(m + n + k)
        x = (torch.randn((m, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        o2 = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        wt = w.t()
        ours = lambda: dga.catlass_dynamic_matmul(x, wt, o)
        vendor = lambda: torch.matmul(x, wt, out=o2)
        ours(); vendor(); torch.cuda.synchronize()
        t = {}
        for rnd in range(3):   # interleaved: both see the same clocks
            for name, fn in (("ours_us", ours), ("vendor_us", vendor)):
                u = sweep.graph_us(fn, 10, replays=3, prewarm_ms=30.0)
                if u:
                    t[name] = min(t.get(name, 1e30), u)
        row = {"shape": [m, n, k], **{a: round(b, 2) for a, b in t.items()}}
        if len(t) == 2:
            row["ours_over_vendor"] = round(t["ours_us"] / t["vendor_us"], 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
