#!/bin/bash
# rocprofv3 kernel-trace summary of the bench step on the data-parallel path with one rank (real RCCL collectives)
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-dp}
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-dist > $R/gpurun_out/prof_$TAG.log 2>&1
find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_${TAG}_kernel_stats.csv \;
tail -2 $R/gpurun_out/prof_$TAG.log | cut -c1-200
