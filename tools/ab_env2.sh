#!/bin/bash
# Same-box A/B of the bench step between two ENVIRONMENTS (each a quoted list of VAR=value), interleaved ROUNDS times:
#   bash tools/ab_env2.sh "A=0 B=0" "A=1 B=1" [ROUNDS] [STEPS]
set -e
E1=$1; E2=$2; ROUNDS=${3:-3}; STEPS=${4:-60}
R=${GRAFT_REPO_ROOT:-.}
cd $R
mkdir -p gpurun_out
: > gpurun_out/ab_env2.log
for r in $(seq 1 $ROUNDS); do
  for e in "$E1" "$E2"; do
    env $e timeout -k 10 200 python3 -u bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-conv-timer > gpurun_out/ab_env_one.json 2> gpurun_out/ab_env_one.err
    python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab_env_one.json').read().strip().splitlines()[-1]); print('[$e]', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a gpurun_out/ab_env2.log
  done
done
