"""Train the tiling-time predictor -- the counterpart of the reference's learned predictor
(/root/reference/get_best_config/model.py:5-30 `TimePredictMLP`: Linear-BatchNorm-ReLU x (64, 32, 16) -> 1, inputs
standardised by a saved scaler; /root/reference/get_best_config/get_best_config.py:281-307 feature rows
(M, N, K, m_tile, n_tile, k_tile), :431-463 greedy selection, :587-621 fallback to the native tiling).

The reference ships no weights (README: model_A2/ and model_A3/ "to be prepared"); here the model is trained on
MI355X sweep records (harness/sweep.py --grid ... --heuristic-raster) and exported, BatchNorm folded, as a plain-text
file that the C++ side (csrc/dga_predictor.cpp) evaluates without Python in the operator path.

  python -m deepgemm_ascend_amd.harness.train_predictor --train DIR [DIR ...] --val DIR --out tuned/predictor_mi355x.txt
"""
from __future__ import annotations

import argparse
import json
import math
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

FEATURES = ["log2_m", "log2_n", "log2_k", "log2_m1", "log2_n1", "stages3", "log2_splitk", "policy1", "policy2",
            "log2_tiles", "log2_rounds", "log2_kb_per_split", "fill_m", "fill_n", "loader_waves", "cold"]
CUS = 256
LDS_PER_CU = 160 * 1024


def stage_bytes(m1, n1):
    """LDS bytes of one pipeline stage (csrc/dga_device_common.hpp GemmCfg): A rows padded to 32, scale slots."""
    a_rows = max(m1, 32)
    return a_rows * 128 + n1 * 128 + ((m1 + 8 + 255) // 256) * 256 * 4


def feature_row(m, n, k, p):
    """One candidate -> the 16 inputs.  Must match csrc/dga_predictor.cpp feature_row() exactly.
    cold: the shape is a short-M weight stream (M <= 256, operands smaller than the Infinity Cache) whose records are timed
    on rotated operand sets (harness/sweep.py --cold) -- another regime than a re-launched shape (HBM round trips, idle CUs
    cost more), which the model is told about rather than left to infer from log2_m."""
    m1, n1, st, sk, pol = p["m1"], p["n1"], p["stages"], p["splitk"], p["policy"]
    tm, tn = -(-m // m1), -(-n // n1)
    tiles = tm * tn * sk
    waves = 8 if (m1 == 256 and n1 == 256) else 4
    lds = stage_bytes(m1, n1) * (3 if st == 3 else 2)
    wg_per_cu = max(1, min(LDS_PER_CU // lds, 2048 // (waves * 64)))
    rounds = -(-tiles // (CUS * wg_per_cu))
    kb = -(-k // 128)
    kb_per_split = -(-kb // sk)
    return [math.log2(m), math.log2(n), math.log2(k), math.log2(m1), math.log2(n1), 1.0 if st == 3 else 0.0,
            math.log2(sk), 1.0 if pol == 1 else 0.0, 1.0 if pol in (2, 6) else 0.0, math.log2(tiles), math.log2(rounds),
            math.log2(kb_per_split), m / (tm * m1), n / (tn * n1), 1.0 if pol in (4, 5) else 0.0,
            1.0 if (m <= 256 and m * k + n * k + 2 * m * n < (256 << 20)) else 0.0]


def load_records(dirs):
    rows = []
    for d in dirs:
        for f in sorted(Path(d).glob("shape_*_rank_*.jsonl")):
            if f.name.endswith("_checkpoint.jsonl"):
                continue
            for line in f.read_text().splitlines():
                r = json.loads(line)
                if r["negative"] or r["time"] <= 0 or r["time"] > 1e8:
                    continue
                # outside the model's candidate space: the quarter-tile tail is a post-pass on the model's pick
                # (apply_tail_split), K % 16 != 0 goes through the padding pass
                if r["parameters"].get("tail") or r["K"] % 16:
                    continue
                rows.append(r)
    return rows


class TimePredictMLP(nn.Module):
    """Same architecture as the reference's model (model.py:5-30): Linear-BatchNorm-ReLU blocks, Kaiming init."""

    def __init__(self, input_dim, hidden_dims=(64, 32, 16)):
        super().__init__()
        layers, prev = [], input_dim
        for h in hidden_dims:
            layers += [nn.Linear(prev, h), nn.BatchNorm1d(h), nn.ReLU()]
            prev = h
        layers.append(nn.Linear(prev, 1))
        self.layers = nn.Sequential(*layers)
        for mod in self.modules():
            if isinstance(mod, nn.Linear):
                nn.init.kaiming_normal_(mod.weight)
                mod.bias.data.fill_(0)

    def forward(self, x):
        return self.layers(x)


def fold(model):
    """[(W, b)] with every BatchNorm folded into the Linear in front of it (eval-mode statistics)."""
    out, mods, i = [], list(model.layers), 0
    while i < len(mods):
        lin = mods[i]
        w, b = lin.weight.detach().double(), lin.bias.detach().double()
        if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d):
            bn = mods[i + 1]
            g = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
            w = w * g[:, None]
            b = (b - bn.running_mean.detach().double()) * g + bn.bias.detach().double()
            i += 3
        else:
            i += 1
        out.append((w.float().numpy(), b.float().numpy()))
    return out


def forward_folded(layers, mean, std, x):
    """numpy statement of what the C++ evaluates (fp32)."""
    h = ((np.asarray(x, np.float32) - mean) / std).astype(np.float32)
    for li, (w, b) in enumerate(layers):
        h = (h @ w.T + b).astype(np.float32)
        if li + 1 < len(layers):
            h = np.maximum(h, 0)
    return h[:, 0]


def export(path, layers, mean, std):
    with open(path, "w") as f:
        f.write("dga-predictor 1\n")
        f.write(f"features {len(mean)} " + " ".join(FEATURES) + "\n")
        f.write("mean " + " ".join(repr(float(v)) for v in mean) + "\n")
        f.write("std " + " ".join(repr(float(v)) for v in std) + "\n")
        f.write(f"layers {len(layers)}\n")
        for w, b in layers:
            f.write(f"layer {w.shape[0]} {w.shape[1]}\n")
            for row in w:
                f.write(" ".join(repr(float(v)) for v in row) + "\n")
            f.write(" ".join(repr(float(v)) for v in b) + "\n")


def by_shape(rows):
    d = {}
    for r in rows:
        d.setdefault((r["M"], r["N"], r["K"]), []).append(r)
    return d


MIN_CANDIDATES = 4       # the reference falls back below 60 candidates of its 16-aligned grid (get_best_config.py:587);
                         # the compiled menu here yields 4..40 per shape
GAIN_THRESHOLD = 0.03    # ... and unless the model promises >= 3 % over the native tiling (:606-616)


def select(pred, rs, native_idx):
    """Greedy pick with the reference's two fallbacks (get_best_config.py:587-621)."""
    best = int(np.argmin(pred))
    if native_idx is None:
        return best
    if len(rs) < MIN_CANDIDATES or pred[best] > (1.0 - GAIN_THRESHOLD) * pred[native_idx]:
        return native_idx
    return best


def evaluate(layers, mean, std, rows, heuristic=None):
    """Per shape: time of the selected candidate / time of the measured-best candidate (1.0 = oracle pick)."""
    ratios, vs_h = [], []
    for (m, n, k), rs in by_shape(rows).items():
        x = np.array([feature_row(m, n, k, r["parameters"]) for r in rs], np.float32)
        pred = np.exp(forward_folded(layers, mean, std, x))
        h = heuristic(m, n, k, rs) if heuristic else None
        pick = rs[select(pred, rs, rs.index(h) if h is not None else None)]
        best = min(r["time"] for r in rs)
        ratios.append(pick["time"] / best)
        if h is not None:
            vs_h.append((pick["time"], h["time"], best))
    return np.array(ratios), vs_h


def heuristic_pick(m, n, k, rs):
    """The record of the candidate the built-in heuristic (select_mi355x, no tuned table) would run."""
    import deepgemm_ascend_amd as dga
    t = dga.select_kernel(m, n, k)
    for r in rs:
        p = r["parameters"]
        pol = {5: 4, 6: 2}.get(t.dispatchPolicyTag, t.dispatchPolicyTag)   # a persistent form is folded into its sibling's record
        if (p["m1"], p["n1"], p["stages"], p["splitk"], p["policy"]) == (t.m1, t.n1, t.stages, t.splitkFactor, pol):
            return r
    return None


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", nargs="+", required=True)
    ap.add_argument("--val", nargs="*", default=[])
    ap.add_argument("--out", default=str(Path(__file__).resolve().parent.parent / "tuned" / "predictor_mi355x.txt"))
    ap.add_argument("--epochs", type=int, default=4000)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    torch.manual_seed(a.seed); np.random.seed(a.seed)
    rows = load_records(a.train)
    x = np.array([feature_row(r["M"], r["N"], r["K"], r["parameters"]) for r in rows], np.float32)
    y = np.log(np.array([r["time"] for r in rows], np.float32))
    mean, std = x.mean(0), x.std(0)
    std[std < 1e-6] = 1.0
    xs = torch.from_numpy((x - mean) / std)
    ys = torch.from_numpy(y)[:, None]
    model = TimePredictMLP(x.shape[1])
    opt = torch.optim.Adam(model.parameters(), lr=3e-3, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, a.epochs)
    model.train()
    for ep in range(a.epochs):
        opt.zero_grad()
        loss = nn.functional.mse_loss(model(xs), ys)
        loss.backward()
        opt.step(); sched.step()
        if ep % 500 == 0:
            print(f"epoch {ep}: mse(log us) {loss.item():.5f}", flush=True)
    model.eval()
    layers = fold(model)
    with torch.no_grad():
        ref = model(xs)[:, 0].numpy()
    got = forward_folded(layers, mean, std, x)
    assert np.allclose(ref, got, atol=2e-4), float(np.abs(ref - got).max())
    export(a.out, layers, mean, std)
    report = {"train_records": len(rows), "train_shapes": len(by_shape(rows)),
              "train_rmse_log": float(np.sqrt(np.mean((got - y) ** 2)))}
    r_train, _ = evaluate(layers, mean, std, rows)
    report["train_pick_over_best"] = {"mean": float(r_train.mean()), "p90": float(np.quantile(r_train, 0.9)), "max": float(r_train.max())}
    if a.val:
        vrows = load_records(a.val)
        r_val, vs_h = evaluate(layers, mean, std, vrows, heuristic_pick)
        report["val_shapes"] = len(by_shape(vrows))
        report["val_pick_over_best"] = {"mean": float(r_val.mean()), "p90": float(np.quantile(r_val, 0.9)), "max": float(r_val.max())}
        if vs_h:
            pick = np.array([v[0] for v in vs_h]); heur = np.array([v[1] for v in vs_h]); best = np.array([v[2] for v in vs_h])
            report["val_heuristic_over_best"] = {"mean": float((heur / best).mean()), "max": float((heur / best).max())}
            report["val_predictor_vs_heuristic_time"] = {"geomean": float(np.exp(np.mean(np.log(pick / heur)))),
                                                         "worst": float((pick / heur).max()), "best": float((pick / heur).min())}
    print(json.dumps(report, indent=1))
    Path(a.out).with_suffix(".report.json").write_text(json.dumps(report, indent=1) + "\n")


if __name__ == "__main__":
    main()
