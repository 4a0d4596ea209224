"""Regret of the bf16-exact policy's selector (dga_tiling_bf16_exact with NO cache row: its cost model and rules) against every
candidate of harness/sweep.py --arith bf16_exact, on the reference's 18 shapes (framework/benchmark/benchmark.py:24-44) and 20 shapes
neither the fit nor the rules have seen (decode rows included; M <= 256 timed cold, on operand sets rotated past the Infinity Cache).
One process; every candidate and the pick by graph replay.  Prints the table, writes the winners that beat the pick by more than
3 % as tag-7 rows to --cache-csv.   Usage: python scripts/r06_bx_regret.py --out gpurun_out/r06/bx_sweep [--cache-csv FILE] [--quick]"""
import argparse
import json
import math
import os
import sys
from pathlib import Path

os.environ["DGA_NO_TUNED_TABLE"] = "1"        # the pick under test is the selector's own, not a row of the shipped table
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402

UNSEEN = [(16, 4096, 7168), (32, 7168, 2048), (48, 18432, 7168), (96, 4096, 7168), (128, 7168, 2048), (200, 5120, 5120),
          (256, 4096, 7168), (384, 7168, 4096), (512, 4096, 7168), (768, 18432, 7168), (1000, 9000, 4000), (1536, 7168, 2048),
          (2048, 2048, 7168), (2304, 4096, 7168), (3000, 3000, 3000), (4096, 7168, 2048), (4096, 2048, 7168), (5120, 5120, 5120),
          (6016, 4096, 4096), (8192, 8192, 2048)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r06/bx_sweep")
    ap.add_argument("--cache-csv", default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--shapes", nargs="*", default=None)
    a = ap.parse_args()
    out_dir = Path(a.out); out_dir.mkdir(parents=True, exist_ok=True)
    shapes = [("reference", tuple(s)) for s in sweep.SHAPE_GROUP] + [("unseen", s) for s in UNSEEN]
    if a.shapes:
        shapes = [("given", tuple(int(x) for x in s.split(","))) for s in a.shapes]
    if a.quick:
        shapes = shapes[::4]
    rows, winners = [], []
    for kind, (m, n, k) in shapes:
        cold = m <= sweep.COLD_MAX_M
        for f in out_dir.glob(f"shape_bx_{m}_{n}_{k}_rank_0*"):      # a fresh table every run
            f.unlink()
        best = sweep.benchmark_shape([m, n, k], out_dir, iters=10, prewarm_s=0.1, cold=cold, arith="bf16_exact")
        if not best:
            continue
        recs = [json.loads(ln) for ln in (out_dir / f"shape_bx_{m}_{n}_{k}_rank_0.jsonl").read_text().splitlines()]
        pick = dga.tiling(m, n, k, policy="bf16_exact")
        key = (int(pick.m1), int(pick.n1), int(pick.kernelSerial), int(pick.splitkFactor), int(pick.build) if pick.build in (8, 10) else 0)
        def key_of(p):
            return (p["m1"], p["n1"], sweep.bx_serial(p), p["splitk"], p.get("build", 0))
        mine = [r for r in recs if not r["negative"] and key_of(r["parameters"]) == key]
        best_us, best_p = best
        if mine:
            pick_us = min(r["time"] for r in mine)
        else:   # a pick outside the candidate grid (a raster, a split): timed here the same way
            a_, sfa, b_, sfb, golden, s_abs = sweep.gen_data(m, n, k)
            o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
            fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a_, sfa), (b_, sfb), o, tiling_=pick, policy="bf16_exact")
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            pick_us = sweep.graph_us(fn, 10, replays=5, prewarm_ms=25.0)
            del a_, sfa, b_, sfb, golden, s_abs, o
        ratio = pick_us / best_us
        rows.append((kind, m, n, k, key, pick_us, key_of(best_p), best_us, ratio, bool(mine)))
        print(f"{kind:9s} {m:5d} x {n:5d} x {k:5d}  pick {key} {pick_us:8.2f} us   best {key_of(best_p)} {best_us:8.2f} us   pick / best {ratio:.3f}"
              f"{'' if mine else '  (pick outside the grid)'}{'  cold' if cold else ''}", flush=True)
        if ratio > 1.03:
            winners.append(((m, n, k), best_p))
        torch.cuda.empty_cache()
    for kind in ("reference", "unseen", "given"):
        rs = [r[8] for r in rows if r[0] == kind]
        if rs:
            print(f"# {kind}: {len(rs)} shapes, pick / best geomean {math.exp(sum(math.log(x) for x in rs) / len(rs)):.4f}, max {max(rs):.3f}, "
                  f"{sum(x > 1.03 for x in rs)} beyond 3 %")
    if a.cache_csv and winners:
        sweep.write_bx_rows(a.cache_csv, winners)
        print(f"# {len(winners)} tag-7 rows written to {a.cache_csv}")


if __name__ == "__main__":
    main()
