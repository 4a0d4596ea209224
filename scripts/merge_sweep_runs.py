"""Merge sweep runs of the same shapes: per candidate the minimum time over the runs (two sweeps of one grid differ by
~12 % rms: clocks, the box), and the persistent form (dispatchPolicyTag 5) folded into its loader-wave sibling (policy 4) --
the selector upgrades 4 to 5 by rule wherever 5 applies, so "loader waves" in the predictor's feature means the faster
of the two (runs that hold policy-5 records go last).  Records of a --cold run replace warm ones of the same candidate.
Usage: python scripts/merge_sweep_runs.py OUT_DIR RUN_DIR [RUN_DIR ...]"""
import json
import sys
from pathlib import Path


def main():
    out = Path(sys.argv[1]); out.mkdir(parents=True, exist_ok=True)
    best = {}
    for d in sys.argv[2:]:
        for f in sorted(Path(d).glob("shape_*_rank_*.jsonl")):
            if f.name.endswith("_checkpoint.jsonl"):
                continue
            for line in f.read_text().splitlines():
                r = json.loads(line)
                p = dict(r["parameters"])
                cold = bool(p.pop("cold_sets", 0))   # timed on operand sets rotated past the Infinity Cache (sweep --cold)
                folded = p.get("policy") in (5, 6)   # persistent forms: loader waves (5 -> 4), continuous pipeline (6 -> 2)
                if folded:
                    p["policy"] = 4 if p["policy"] == 5 else 2
                key = (f.name, json.dumps(p, sort_keys=True))
                if folded and key not in best:   # no loader-wave record of that candidate in the earlier runs: not a candidate
                    continue                     # of the predictor's menu (dga_predictor.cpp), leave it out
                bad = r["negative"] or r["time"] <= 0
                cur = best.get(key)
                # a cold timing replaces a warm one of the same candidate whatever the two say (short-M shapes are tuned for
                # weights that come from HBM); among timings of one kind the faster wins
                if cur is not None and cur.get("_cold") and not cold:
                    continue
                if cur is None or (cold and not cur.get("_cold")) or (not bad and (cur["negative"] or r["time"] < cur["time"])):
                    best[key] = dict(r, parameters=p, _cold=cold)
    files = {}
    for (name, _), r in best.items():
        files.setdefault(name, []).append(r)
    for name, recs in files.items():
        recs.sort(key=lambda r: r["idx"])
        for r in recs:
            if r.pop("_cold", False):
                r["parameters"]["cold"] = 1
        (out / name).write_text("".join(json.dumps(r) + "\n" for r in recs))
    print(f"{len(best)} records over {len(files)} shapes -> {out}")


if __name__ == "__main__":
    main()
