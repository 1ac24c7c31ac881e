#!/bin/bash
# rocprofv3 kernel trace (ordered launches) + stats of the bench step; writes gpurun_out/prof_<tag>_kernel_{stats,trace}.csv
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-trace}
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_${TAG}_kernel_stats.csv \;
find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" -exec cp {} $R/gpurun_out/prof_${TAG}_kernel_trace.csv \;
rm -rf $R/gpurun_out/prof_$TAG
tail -2 $R/gpurun_out/prof_$TAG.log | cut -c1-300
