#!/bin/bash
# The measurement pass whose outputs are committed under profiles/ (run on the GPU box through gpurun):
#   bench line, rocprofv3 kernel stats of the same command, PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately).
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
python3 $R/bench.py --steps 50 --warmup 10 --widen --no-cpu-baseline --no-grouped > $O/bench_widen.json 2>> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r01 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
echo "stats done"
for w in dense grouped; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${w}_$c -o p -- python3 $R/scripts/prof_$w.py > $O/pmc_${w}_$c.log 2>&1
    echo "pmc $w $c done"
  done
done
find $O -name "*.csv" | head -30
