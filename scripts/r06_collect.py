"""Condense the outputs of scripts/r06_measure.sh into the files committed under profiles/ (r06_*)."""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

src, dst = Path(sys.argv[1]), Path(sys.argv[2])
dst.mkdir(parents=True, exist_ok=True)


def find(d, suffix):
    hits = sorted(Path(d).rglob("*" + suffix))
    return hits[0] if hits else None


def counters(d, match, skip):
    """mean per launch of every counter of the kernels whose name contains `match`, skipping the first `skip` launches"""
    f = find(d, "_counter_collection.csv")
    if not f:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if match in r["Kernel_Name"]:
            acc[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    out = {}
    for name, per in acc.items():
        vals = [v for _, v in sorted(per.items())][skip:]
        out[name] = sum(vals) / max(1, len(vals))
    kt = find(d, "_kernel_trace.csv")
    if kt:
        ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if match in r["Kernel_Name"]][skip:]
        out["kernel_us_under_profiler"] = sum(ds) / max(1, len(ds))
        out["launches_averaged"] = len(ds)
        names = collections.Counter(r["Kernel_Name"] for r in csv.DictReader(open(kt)) if match in r["Kernel_Name"])
        out["kernel"] = names.most_common(1)[0][0] if names else None
    return out


for name in ("bench.json", "bench_detail.json", "bench_under_rocprof.json"):
    if (src / name).exists():
        shutil.copy(src / name, dst / ("r06_" + name))
st = find(src / "stats", "_kernel_stats.csv")
if st:
    shutil.copy(st, dst / "r06_kernel_stats.csv")

# the same trace grouped by launch configuration (one kernel symbol serves several shapes of the bench)
kt = find(src / "stats", "_kernel_trace.csv")
if kt:
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        if "dga::" in r["Kernel_Name"]:
            groups[(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(dst / "r06_kernel_stats_by_grid.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Grid_Size_X", "Workgroup_Size_X", "Calls", "AverageNs", "MinNs", "MaxNs"])
        for (name, grid, wg), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, grid, wg, len(v), round(sum(v) / len(v), 1), min(v), max(v)])

# configs[1] isolated: the dense kernel's launches after the first 400 (sustained clocks), one kernel-trace-only pass per policy
rows = []
for pol in ("bf16_exact", "fast"):
    kt = find(src / f"dense_{pol}", "_kernel_trace.csv")
    if not kt:
        continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        if "dga::" in r["Kernel_Name"]:
            per[(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for (name, grid, wg), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        v = v[400:] if len(v) > 450 else v
        rows.append([pol, name, grid, wg, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
if rows:
    with open(dst / "r06_dense_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Policy", "Name", "Grid_Size_X", "Workgroup_Size_X", "Calls(after 400 warm)", "AverageNs", "MinNs", "MaxNs"])
        w.writerows(rows)

traffic = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (TCC slot limit), mean per launch of the "
                   "fp8 GEMM kernel after 400 warm launches (dense) / 5 (grouped); gfx950 correction per MI355X_MICROARCH.md "
                   "section HBM: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2; units are KB -> x1024. "
                   "Infinity-Cache hits are included (fabric-side counter), so this is an upper bound on HBM bytes.  Keys without "
                   "a suffix are the fast policy's kernels (as in earlier rounds); *_bf16_exact = the operator's default policy.",
           "command": "scripts/r06_measure.sh"}
for key, tag, skip, alg in (("dense", "dense_fast", 400, 67637248), ("dense_bf16_exact", "dense_bf16_exact", 400, 67637248),
                            ("dsv3_prefill", "prefill_fast", 400, 61738496), ("dsv3_prefill_bf16_exact", "prefill_bf16_exact", 400, 61738496),
                            ("grouped", "grouped_fast", 5, 4135452672), ("grouped_bf16_exact", "grouped_bf16_exact", 5, 4135452672)):
    f = counters(src / f"pmc_{tag}_FETCH_SIZE", "gemm_fp8", skip)
    w = counters(src / f"pmc_{tag}_WRITE_SIZE", "gemm_fp8", skip)
    if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
        traffic[key] = {"FETCH_SIZE_KB": round(f["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(w["WRITE_SIZE"], 1),
                        "traffic_bytes": int(f["FETCH_SIZE"] * 2 * 1024 + w["WRITE_SIZE"] * 1024), "algorithmic_bytes": alg,
                        "kernel_us_under_profiler": round(f.get("kernel_us_under_profiler", 0), 2), "kernel": f.get("kernel")}
(dst / "r06_traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")

util = {}
for pol in ("bf16_exact", "fast"):
    m = counters(src / f"pmc_dense_mfma_{pol}", "gemm_fp8", 400)
    if m:
        busy, gui = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), m.get("GRBM_GUI_ACTIVE", 0)
        us = m.get("kernel_us_under_profiler", 0)
        util[pol] = {"kernel": m.get("kernel"), "counters": {k: round(v, 1) for k, v in m.items() if isinstance(v, float)},
                     "mfma_busy_cycles_per_simd": round(busy / 1024, 1), "kernel_cycles_per_xcd": round(gui / 8, 1),
                     "mfma_pipe_busy_fraction_of_kernel_cycles": round((busy / 1024) / (gui / 8), 4) if gui else None,
                     "effective_clock_ghz": round(gui / 8 / us / 1e3, 3) if us else None}
if util:
    util["command"] = ("rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 scripts/prof_dense.py "
                       "4096 4096 4096 600 --policy <policy>  (means over launches 401..600: sustained clocks)")
    (dst / "r06_mfma_util.json").write_text(json.dumps(util, indent=1) + "\n")
rows = []
for mk in ("full", "random"):
    for pol in ("bf16_exact", "fast"):
        kt = find(src / f"grouped_{mk}_{pol}", "_kernel_trace.csv")
        if not kt:
            continue
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(kt)):
            if "dga::" in r["Kernel_Name"]:
                per[(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (name, grid, wg), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            v = v[10:] if len(v) > 20 else v
            rows.append([mk, pol, name, grid, wg, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
if rows:
    with open(dst / "r06_grouped_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Mask", "Policy", "Name", "Grid_Size_X", "Workgroup_Size_X", "Calls(after 10 warm)", "AverageNs", "MinNs", "MaxNs"])
        w.writerows(rows)
for name in ("grouped_masks.txt", "grouped_stamps.txt"):
    if (src / name).exists():
        shutil.copy(src / name, dst / ("r06_" + name))
print("collected into", dst)
