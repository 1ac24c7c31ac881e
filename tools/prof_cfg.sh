#!/bin/bash
# rocprofv3 kernel stats of tools/bench_configs.py: prof_cfg.sh <tag> <args...>
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/tools/bench_configs.py "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_${TAG}_kernel_stats.csv \;
rm -rf $R/gpurun_out/prof_$TAG
grep '^{' $R/gpurun_out/prof_$TAG.log | cut -c1-400
