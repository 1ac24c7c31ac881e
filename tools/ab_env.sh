#!/bin/bash
# Same-box A/B of the bench step over values of ONE environment variable, interleaved, ROUNDS times:
#   bash tools/ab_env.sh VAR "v1 v2 v3" [ROUNDS] [STEPS]
# prints "VAR=v ms_per_step" per run (gpurun_out/ab_env.log keeps them)
set -e
VAR=$1; VALS=$2; ROUNDS=${3:-2}; STEPS=${4:-60}
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out
: > gpurun_out/ab_env.log
for r in $(seq 1 $ROUNDS); do
  for v in $VALS; do
    export $VAR=$v
    timeout -k 10 200 python3 -u bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-conv-timer > gpurun_out/ab_env_one.json 2> gpurun_out/ab_env_one.err
    python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab_env_one.json').read().strip().splitlines()[-1]); print('$VAR=$v', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a gpurun_out/ab_env.log
  done
done
