#!/bin/bash
# HBM traffic of the bench step from PMC counters: FETCH_SIZE and WRITE_SIZE in separate runs (TCC slots),
# no trace domains.  Run on the GPU box: bash tools/pmc_bench.sh ; results under gpurun_out/pmc_bench/.
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
export UNET_AMD_OVERLAP_WGRAD=0     # per-kernel durations without cross-stream queueing (the product default overlaps the weight gradients)
cd /tmp
O=$R/gpurun_out/pmc_bench
rm -rf $O; mkdir -p $O
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conv-timer > $O/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conv-timer > $O/write.log 2>&1
python3 $R/tools/pmc_traffic.py $O > $O/summary.txt
tail -25 $O/summary.txt
