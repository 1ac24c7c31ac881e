#!/bin/bash
# rocprofv3 kernel-trace summary of the bench step (run on the GPU box): tools/prof_bench.sh <tag>
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-cur}
export TMPDIR=/tmp
export UNET_AMD_OVERLAP_WGRAD=0     # per-kernel durations without cross-stream queueing (the product default overlaps the weight gradients)
cd /tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_${TAG}_kernel_stats.csv \;
tail -2 $R/gpurun_out/prof_$TAG.log
