#!/bin/bash
# HBM traffic of a tools/bench_configs.py workload from PMC counters (FETCH_SIZE and WRITE_SIZE in separate passes, no trace
# domains): bash tools/pmc_cfg.sh <tag> <kernel-filter> <bench_configs args...>; results under gpurun_out/pmc_<tag>/.
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; FILTER=$2; shift 2
export TMPDIR=/tmp
cd /tmp
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/bench_configs.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-timer > $O/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/bench_configs.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-timer > $O/write.log 2>&1
python3 $R/tools/pmc_traffic.py $O "$FILTER" > $O/summary.txt
tail -3 $O/summary.txt
