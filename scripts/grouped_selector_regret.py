"""The grouped selector on shapes it was never tuned on: every legal build of the menu timed on a masked / contiguous grouped
problem (harness/sweep.py benchmark_grouped: full mask, correctness-gated) against the tiling dga_tiling() names for it.
Usage: python scripts/grouped_selector_regret.py [layout,groups,rows,n,k ...]"""
import json
import math
import sys
import tempfile
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402

SHAPES = ([("masked", g, mm, n, k) for (n, k) in ((5120, 5120), (3072, 8192), (1536, 4096)) for (g, mm) in ((64, 16), (64, 128), (16, 64), (128, 32))] +
          [("contiguous", g, r, n, k) for (n, k) in ((5120, 5120), (2048, 4096)) for (g, r) in ((16, 128), (4, 1024), (2, 4096))])


def main():
    global SHAPES
    if len(sys.argv) > 1:   # layout,groups,rows,n,k ...
        SHAPES = [(s.split(",")[0],) + tuple(int(x) for x in s.split(",")[1:]) for s in sys.argv[1:]]
    ratios = []
    out_dir = Path(tempfile.mkdtemp())
    for shape in SHAPES:
        layout, groups, rows, n, k = shape
        prob, best = sweep.benchmark_grouped(shape, out_dir, iters=10)
        recs = [json.loads(l) for l in open(out_dir / f"shape_{layout}_{groups}x{rows}_{n}_{k}_rank_0.jsonl")]
        if layout == "masked":
            t = dga.tiling(rows, n, k, groups=groups, expected_m=rows)
        else:
            t = dga.tiling(groups * rows, n, k, groups=groups, contiguous=True)
        key = (int(t.m1), int(t.n1), int(t.stages), int(t.dispatchPolicyTag))
        mine = [r["time"] for r in recs if not r["negative"] and
                (r["parameters"]["m1"], r["parameters"]["n1"], r["parameters"]["stages"], r["parameters"]["policy"]) == key]
        if not mine or not best:
            print(json.dumps({"shape": list(shape), "pick": key, "note": "no record of the pick"}), flush=True)
            continue
        ratios.append(min(mine) / best[0])
        print(json.dumps({"shape": list(shape), "pick": key, "pick_us": round(min(mine), 1), "best_us": round(best[0], 1),
                          "best": [best[1][c] for c in ("m1", "n1", "stages", "policy")], "ratio": round(ratios[-1], 3),
                          "top": sorted(((round(r["time"], 1), r["parameters"]["m1"], r["parameters"]["n1"], r["parameters"]["stages"],
                                          r["parameters"]["policy"]) for r in recs if not r["negative"]))[:6]}), flush=True)
    print("shapes", len(ratios), "geomean", round(math.exp(sum(math.log(r) for r in ratios) / max(1, len(ratios))), 4), "max", round(max(ratios), 3))


if __name__ == "__main__":
    main()
