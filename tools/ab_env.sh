#!/bin/bash
# same-box A/B of the conv microbenchmark between two settings of one environment variable:
#   tools/ab_env.sh VAR valueA valueB layer...
VAR=$1; A=$2; B=$3; shift 3
for l in "$@"; do
  for v in $A $B; do
    echo -n "$VAR=$v "
    env $VAR=$v timeout -k 10 200 python -u tools/conv_microbench.py --iters 20 --only $l 2>&1 | grep "\->" | head -1 || exit 1
  done
done
