"""From a scripts/r06_bx_regret.py output directory: per shape, the best one-launch decode split-K candidate (build 10) against the best
of every other candidate of the bf16-exact menu, both timed by the sweep's protocol (cold for M <= 256).  Usage: python
scripts/r06_decode_cold_table.py gpurun_out/r06/bx_decode_wide"""
import glob, json, re, sys
rows = []
for f in sorted(glob.glob(sys.argv[1] + "/shape_bx_*_rank_0.jsonl")):
    recs = [json.loads(l) for l in open(f)]
    ok = [r for r in recs if not r["negative"]]
    dsk = [r for r in ok if r["parameters"].get("build") == 10]
    oth = [r for r in ok if r["parameters"].get("build") != 10]
    if not dsk or not oth:
        continue
    bd, bo = min(dsk, key=lambda r: r["time"]), min(oth, key=lambda r: r["time"])
    m, n, k = (int(x) for x in re.search(r"shape_bx_(\d+)_(\d+)_(\d+)", f).groups())
    p = bo["parameters"]
    tiles, kb = -(-m // 64) * -(-n // 128), -(-k // 128)
    smax = max(1, min(8, 256 // tiles, kb // 4))
    rows.append((m, n, k, tiles, kb / (2 * smax), bd["parameters"]["splitk"], bd["time"], f"{p['m1']}x{p['n1']} s{p['splitk']}{' wsk' if p.get('wsk') else ''}", bo["time"]))
for r in sorted(rows):
    print(f"{r[0]:4d} x {r[1]:5d} x {r[2]:5d}  tiles {r[3]:3d}  k blocks per group at smax {r[4]:5.1f}  decode build s{r[5]} {r[6]:7.2f} us   best other {r[7]:14s} {r[8]:7.2f} us   ratio {r[6] / r[8]:.3f}")
