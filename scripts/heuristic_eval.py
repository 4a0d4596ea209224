"""Offline score of the built-in tile heuristic against the sweep records (no GPU): time of the heuristic's candidate /
time of the measured-best candidate per shape."""
import os, sys
os.environ["DGA_NO_TUNED_TABLE"] = "1"; os.environ["DGA_NO_PREDICTOR"] = "1"
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import train_predictor as tp

dirs = sys.argv[1:] or ["profiles/r01_predictor/val", "profiles/r01_predictor/val2", "profiles/r01_predictor/train", "profiles/r01_predictor/train2"]
rows = tp.load_records(dirs)
ratios, missing, worst = [], 0, []
for (m, n, k), rs in tp.by_shape(rows).items():
    h = tp.heuristic_pick(m, n, k, rs)
    if h is None:
        missing += 1
        continue
    best = min(r["time"] for r in rs)
    ratios.append(h["time"] / best)
    worst.append((h["time"] / best, (m, n, k), h["parameters"], min(rs, key=lambda r: r["time"])["parameters"]))
r = np.array(ratios)
print(f"shapes {len(r)} (+{missing} whose heuristic pick was not swept): mean {r.mean():.4f}  geomean {np.exp(np.log(r).mean()):.4f}  p90 {np.quantile(r, 0.9):.3f}  max {r.max():.3f}")
for w in sorted(worst, reverse=True, key=lambda x: x[0])[:12]:
    print(f"  {w[0]:.2f}x {w[1]}  heuristic {w[2]}  best {w[3]}")
