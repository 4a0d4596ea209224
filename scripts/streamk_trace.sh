#!/bin/bash
# kernel durations of the Stream-K kernel and of the selector's pick, from the kernel trace (gpurun_out/sk)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sk
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in "4096 4096 4096" "1024 18432 7168"; do
  tag=$(echo $s | tr ' ' x)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o t -- python3 $R/scripts/streamk_trace.py $s > $O/$tag.log 2>&1
  find $O/$tag -name "*kernel_stats.csv" -exec cut -c1-220 {} \; | head -8
done
