"""CPU: the dense selector against the committed device-timed sweep records (profiles/r03_predictor/val.tar.gz: the 120 shapes
HELD OUT of the cost-model fit, every (tile, split-K, stages, policy) candidate timed by graph replay on an MI355X).  The time
recorded for the candidate the selector names, over the best recorded candidate of the shape: a regression guard for
csrc/dga_tiling.cpp's cost model and build rules (round 3: 1.024 geomean, max 1.23; the tile-first rule it replaced: 1.24 / 2.45).
Same for the bf16-exact policy's tiling on its own sweep (bf16_exact_sweep.tar.gz, val/: 1.018, max 1.32)."""
import json
import math
import tarfile
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REC = ROOT / "profiles" / "r03_predictor"


def _load(tar_name, member_prefix, tmp_path, key_of):
    with tarfile.open(REC / tar_name) as tf:
        tf.extractall(tmp_path)
    shapes = {}
    for f in sorted((tmp_path / member_prefix).glob("shape_*_rank_*.jsonl")):
        for line in f.read_text().splitlines():
            r = json.loads(line)
            if r["negative"] or r["time"] <= 0:
                continue
            cur = shapes.setdefault((r["M"], r["N"], r["K"]), {})
            k = key_of(r["parameters"])
            cur[k] = min(cur.get(k, 1e30), r["time"])
    return shapes


def _stats(reg):
    return math.exp(sum(map(math.log, reg)) / len(reg)), max(reg)


def test_dense_selector_regret_on_the_held_out_sweep(dga, tmp_path, monkeypatch):
    monkeypatch.setenv("DGA_NO_PREDICTOR", "1")
    fold = lambda p: {5: 4, 6: 2}.get(p, p)
    shapes = _load("val.tar.gz", "pred3_val", tmp_path,
                   lambda p: (p["m1"], p["n1"], p["stages"], p["splitk"], fold(p["policy"]), bool(p.get("tail"))))
    assert len(shapes) >= 100
    reg, missing = [], 0
    for (m, n, k), cs in shapes.items():
        t = dga.select_kernel(m, n, k)
        if t.kernelSerial == 6:   # the workgroup split-K (decode rows): not a candidate of this sweep's tile menu; its own cold
            continue              # sweep against the tile kernels is profiles/r04_sweep_wskd/table.txt
        key = (t.m1, t.n1, 3 if t.stages == 3 else 2, max(1, t.splitkFactor), fold(t.dispatchPolicyTag), t.kernelSerial == 5)
        if key not in cs:
            missing += 1
            continue
        reg.append(cs[key] / min(cs.values()))
    geo, worst = _stats(reg)
    assert missing <= 6, missing                       # the selector names candidates the sweep covers
    assert geo <= 1.04 and worst <= 1.35, (geo, worst)


def test_bf16_exact_tiling_regret_on_its_held_out_sweep(dga, tmp_path):
    shapes = _load("bf16_exact_sweep.tar.gz", "val", tmp_path, lambda p: (p["m1"], p["n1"], p["splitk"]))
    assert len(shapes) >= 100
    reg, missing = [], 0
    for (m, n, k), cs in shapes.items():
        t = dga.tiling(m, n, k, policy="bf16_exact")
        if t.kernelSerial == 6:   # decode rows on the workgroup split-K: outside this sweep's menu (profiles/r04_wskd_cold_bf16x.txt)
            continue
        key = (t.m1, t.n1, max(1, t.splitkFactor))
        if key not in cs:
            missing += 1
            continue
        reg.append(cs[key] / min(cs.values()))
    geo, worst = _stats(reg)
    assert missing <= 10, missing
    assert geo <= 1.04 and worst <= 1.4, (geo, worst)
