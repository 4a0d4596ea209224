"""16-bit operator (bf16) at more than 128 rows: the planned launch against forced plans -- with / without the sub-tile tail, the 4- and
8-wave builds of the 128 x 128 tile -- and the vendor GEMM library (yardstick), warm, device time by graph replay.
Usage: python scripts/op16_tail_ab.py"""
import json
import os
os.environ.setdefault("DGA_B16_DEV", "1")   # the 16-bit operators read their development switches per call only when told so (dga_b16.hip)

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402

SHAPES = [(1024, 18432, 7168), (5119, 6997, 9901), (1024, 4096, 7168), (2048, 4096, 7168), (4096, 4096, 4096), (3511, 6151, 8191),
          (1279, 5003, 7681), (2304, 8192, 4096), (1536, 12288, 5120), (4608, 4096, 7168), (3072, 6144, 4096), (8192, 4608, 4096),
          (512, 7168, 4096), (768, 8192, 8192), (1024, 2048, 7168), (256, 7168, 7168), (384, 16384, 4096), (640, 5120, 5120)]
PLANS = [None, "256,256,1,0", "256,256,1,128", "256,256,1,64", "256,256,1,32", "128,256,1,0", "128,256,2,0", "128,256,1,64", "128,128,1,0",
         "128,128,1,0,1", "128,128,2,0,1"]   # (fifth field 1: the 8-wave three-stage build of the 128 x 128 tile)


def main():
    for (m, n, k) in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(m + n + k)
        x = (torch.randn((m, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        fn = lambda: dga.catlass_dynamic_matmul(x, w.t(), o)
        row = {"shape": [m, n, k]}
        for rnd in range(2):
            for plan in PLANS:
                if plan:
                    os.environ["DGA_B16_PLAN"] = plan
                else:
                    os.environ.pop("DGA_B16_PLAN", None)
                u = sweep.graph_us(fn, 10, replays=3, prewarm_ms=30.0)
                key = plan or "auto"
                row[key] = round(min(row.get(key, 1e30), u), 1)
        os.environ.pop("DGA_B16_PLAN", None)
        v = lambda: torch.matmul(x, w.t(), out=o)
        row["vendor"] = round(min(sweep.graph_us(v, 10, replays=3, prewarm_ms=30.0) for _ in range(2)), 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
