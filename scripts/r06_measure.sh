#!/bin/bash
# Round-6 measurement pass (run on the GPU box through gpurun); outputs land in gpurun_out/r06 and the summaries are copied
# into profiles/ by scripts/r06_collect.py.  PMC passes are separate runs with --kernel-trace only (the pool refuses --pmc
# together with the runtime / sys trace domains), FETCH_SIZE and WRITE_SIZE in their own passes (TCC slots).
# The headline policy is the operator's default, bf16_exact (in contract); "fast" = the opt-in fp8-instruction policy.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 200 --warmup 20 --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err
echo "bench done"
# the command the driver runs, under the kernel trace (same steps / warmup as the driver's own call)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r06 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic --detail-out $O/bench_under_rocprof_detail.json > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
echo "stats done"
# the headline kernel on its own in the trace (the bench's trace mixes it with the grouped launches of the same grid size)
for pol in bf16_exact fast; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense_$pol -o d -- python3 $R/scripts/prof_dense.py 4096 4096 4096 600 --policy $pol > $O/dense_$pol.log 2>&1
done
echo "dense traces done"
for c in FETCH_SIZE WRITE_SIZE; do
  for pol in bf16_exact fast; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_dense_${pol}_$c -o p -- python3 $R/scripts/prof_dense.py 4096 4096 4096 420 --policy $pol > $O/pmc_dense_${pol}_$c.log 2>&1
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_prefill_${pol}_$c -o p -- python3 $R/scripts/prof_dense.py 4096 2048 7168 420 --policy $pol > $O/pmc_prefill_${pol}_$c.log 2>&1
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_grouped_${pol}_$c -o p -- python3 $R/scripts/prof_grouped.py 30 full $pol > $O/pmc_grouped_${pol}_$c.log 2>&1
  done
  echo "pmc $c done"
done
# matrix-pipe busy / active cycles of the two dense kernels at sustained clocks (>= 400 warm launches in the same process)
for pol in bf16_exact fast; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_dense_mfma_$pol -o p -- python3 $R/scripts/prof_dense.py 4096 4096 4096 600 --policy $pol > $O/pmc_dense_mfma_$pol.log 2>&1
done
echo "pmc mfma done"
# configs[3] on its own in the trace: one kernel-trace-only pass per mask and policy
for mk in full random; do
  for pol in bf16_exact fast; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/grouped_${mk}_$pol -o g -- python3 $R/scripts/prof_grouped.py 60 $mk $pol > $O/grouped_${mk}_$pol.log 2>&1
  done
done
echo "grouped traces done"
# configs[3] by mask class, in-contract default against the fast policy, one process (no profiler)
python3 $R/scripts/r06_grouped_masks.py > $O/grouped_masks.txt 2>&1
# ... and where the grouped kernel's waves spend a k block (stamped build)
( cd $R/scripts/ubench && for r in 128 96 80 64 48 16 -7; do ./stamp_grouped_bx 256 2048 7168 $r 60; done ) > $O/grouped_stamps.txt 2>&1
echo "grouped tables done"
python3 $R/scripts/r06_collect.py $O $O/summary
ls $O/summary
# the decode build on its own in the trace (eager launches: the durations are the kernel's, the launch gaps are not in them)
for shp in "128 4096 7168" "64 4096 7168" "64 7168 18432" "256 4096 7168"; do
  tag=$(echo $shp | tr ' ' 'x')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/decode_$tag -o d -- python3 $R/scripts/prof_dense.py $shp 300 --policy bf16_exact > $O/decode_$tag.log 2>&1
done
python3 - <<PY
import collections, csv, glob
rows = []
for d in sorted(glob.glob("$O/decode_*/")):
    for f in glob.glob(d + "**/*_kernel_trace.csv", recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "dga::" in r["Kernel_Name"]:
                per[(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (name, grid, wg), v in per.items():
            v = v[100:] if len(v) > 150 else v
            rows.append([d.rstrip("/").split("decode_")[-1], name, grid, wg, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
with open("$O/summary/r06_decode_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Shape", "Name", "Grid_Size_X", "Workgroup_Size_X", "Calls(after 100 warm)", "AverageNs", "MinNs", "MaxNs"])
    w.writerows(rows)
PY
echo "decode traces done"
