#!/bin/bash
# Round-4 measurement pass (run on the GPU box through gpurun); outputs land in gpurun_out/r04 and the summaries are
# copied into profiles/ by scripts/r04_collect.py.  PMC passes are separate runs with --kernel-trace only (the pool
# refuses --pmc together with the runtime / sys trace domains), FETCH_SIZE and WRITE_SIZE in their own passes (TCC slots).
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r04 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-live-traffic > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
echo "stats done"
# traffic: dense configs[1], configs[2], grouped configs[3]; 400 warm launches first so that the counters are read at
# sustained clocks
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_dense_$c -o p -- python3 $R/scripts/prof_dense.py 4096 4096 4096 420 > $O/pmc_dense_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_prefill_$c -o p -- python3 $R/scripts/prof_dense.py 4096 2048 7168 420 > $O/pmc_prefill_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_grouped_$c -o p -- python3 $R/scripts/prof_grouped.py 30 > $O/pmc_grouped_$c.log 2>&1
  # the bf16-exact policy's kernel on configs[1] ($DGA_BF16_EXACT=1 forces dispatchPolicyTag 7 for every fp8 call of the process)
  DGA_BF16_EXACT=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_bf16x_$c -o p -- python3 $R/scripts/prof_dense.py 4096 4096 4096 420 > $O/pmc_bf16x_$c.log 2>&1
  echo "pmc $c done"
done
# matrix-pipe busy / active cycles of the dense kernel at sustained clocks (>= 400 warm launches in the same process)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_dense_mfma -o p -- python3 $R/scripts/prof_dense.py 4096 4096 4096 600 > $O/pmc_dense_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_grouped_occ -o p -- python3 $R/scripts/prof_grouped.py 30 > $O/pmc_grouped_occ.log 2>&1
echo "pmc mfma done"
# configs[3] on its own in the trace: one kernel-trace-only pass per mask (the bench's trace mixes every mask and the sampled-expert
# parity launches under one grid size)
for mk in full random; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/grouped_$mk -o g -- python3 $R/scripts/prof_grouped.py 60 $mk > $O/grouped_$mk.log 2>&1
done
echo "grouped traces done"
python3 $R/scripts/policy_perf.py > $O/policy_perf.txt 2>&1
echo "policy table done"
python3 $R/scripts/r04_collect.py $O $O/summary
ls $O/summary
