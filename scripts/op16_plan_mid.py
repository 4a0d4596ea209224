"""16-bit operator (bf16) at more than 128 rows, shapes the planner's constants were not set on: the planned launch against every plan
of the menu (tile, split-K, sub-tile tail, the 128x128 tile's two builds), warm, device time by graph replay.
Usage: python scripts/op16_plan_mid.py"""
import json
import math
import os
os.environ.setdefault("DGA_B16_DEV", "1")   # the 16-bit operators read their development switches per call only when told so (dga_b16.hip)

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402

SHAPES = [(512, 5120, 5120), (1024, 13824, 5120), (2048, 5120, 13824), (768, 14336, 4096), (1536, 3072, 8192), (384, 8192, 8192), (256, 6144, 4096),
          (640, 11008, 4096), (1280, 2560, 10240), (3072, 5120, 5120), (896, 28672, 4096), (448, 16384, 2048), (1792, 7168, 4608), (2560, 3584, 7168),
          (320, 12288, 6144), (4096, 5120, 2560), (1152, 9216, 4608), (736, 6656, 8192), (192, 4096, 4096), (160, 14336, 4096)]
PLANS = ([f"256,256,{s}" for s in (1, 2, 3, 4)] + ["256,256,1,128", "256,256,1,64", "256,256,1,32"] + [f"128,256,{s}" for s in (1, 2, 3, 4, 6)] +
         ["128,256,1,64", "128,256,1,32"] + [f"128,128,{s}" for s in (1, 2, 3)] + [f"128,128,{s},0,1" for s in (1, 2, 3, 4)] + [f"64,128,{s}" for s in (1, 2, 4)])


def main():
    reg = []
    for (m, n, k) in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(m + n + k)
        x = (torch.randn((m, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        fn = lambda: dga.catlass_dynamic_matmul(x, w.t(), o)
        t = {}
        for rnd in range(2):
            for plan in [None] + PLANS:
                if plan:
                    if plan.count(",") == 2 and int(plan.split(",")[2]) > 1 and (k // 64) // int(plan.split(",")[2]) < 8:
                        continue
                    os.environ["DGA_B16_PLAN"] = plan
                else:
                    os.environ.pop("DGA_B16_PLAN", None)
                u = sweep.graph_us(fn, 10, replays=3, prewarm_ms=20.0)
                t[plan or "auto"] = min(t.get(plan or "auto", 1e30), u)
        os.environ.pop("DGA_B16_PLAN", None)
        best = min((p for p in t if p != "auto"), key=t.get)
        reg.append(t["auto"] / t[best])
        print(json.dumps({"shape": [m, n, k], "auto_us": round(t["auto"], 1), "best": best, "best_us": round(t[best], 1), "auto_over_best": round(reg[-1], 3),
                          "plans": {p: round(u, 1) for p, u in sorted(t.items(), key=lambda kv: kv[1])[:5]}}), flush=True)
    print("shapes", len(reg), "geomean", round(math.exp(sum(math.log(r) for r in reg) / len(reg)), 4), "max", round(max(reg), 3))


if __name__ == "__main__":
    main()
