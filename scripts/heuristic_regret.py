"""Regret of the selector without the learned predictor on sweep records: per shape, the recorded time of the candidate the
heuristic names (its tile, split-K, stage count and policy; persistent forms folded into their one-tile siblings) over the best
recorded candidate.  usage: DGA_NO_PREDICTOR=1 python scripts/heuristic_regret.py DIR [DIR ...]"""
import glob, json, math, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("DGA_NO_PREDICTOR", "1")
import deepgemm_ascend_amd as dga

shapes = {}
for d in sys.argv[1:]:
    for f in glob.glob(str(Path(d) / "shape_*_rank_*.jsonl")):
        if f.endswith("_checkpoint.jsonl"):
            continue
        for line in open(f):
            r = json.loads(line)
            if r["negative"] or r["time"] <= 0:
                continue
            p = r["parameters"]
            pol = {5: 4, 6: 2}.get(p["policy"], p["policy"])
            key = (p["m1"], p["n1"], p["stages"], p["splitk"], pol, bool(p.get("tail")))
            cur = shapes.setdefault((r["M"], r["N"], r["K"]), {})
            cur[key] = min(cur.get(key, 1e30), r["time"])
rows, missing, wsk = [], 0, 0
for (m, n, k), cs in shapes.items():
    t = dga.select_kernel(m, n, k)
    if t.kernelSerial == 6:   # the one-launch workgroup split-K: not a candidate of these records
        wsk += 1
        continue
    pol = {5: 4, 6: 2}.get(t.dispatchPolicyTag, t.dispatchPolicyTag)
    key = (t.m1, t.n1, 3 if t.stages == 3 else 2, max(1, t.splitkFactor), pol, t.kernelSerial == 5)
    best = min(cs.values())
    if key not in cs:
        alt = [v for kk, v in cs.items() if kk[:2] == key[:2] and kk[3] == key[3]]
        if not alt:
            missing += 1
            continue
        rows.append((min(alt) / best, (m, n, k), key, min(alt), best, "~"))
    else:
        rows.append((cs[key] / best, (m, n, k), key, cs[key], best, ""))
reg = [r[0] for r in rows]
print(f"{len(rows)} shapes ({missing} with no record of the pick): pick / best geomean {math.exp(sum(map(math.log, reg)) / len(reg)):.4f} "
      f"mean {sum(reg) / len(reg):.4f} p90 {sorted(reg)[int(0.9 * len(reg))]:.3f} max {max(reg):.3f}")
for r in sorted(rows, reverse=True)[:15]:
    print("  %.2f %s pick %s %.1f best %.1f %s" % r)
