#!/bin/bash
# same-box A/B of the conv microbenchmark between two libraries: tools/ab_libs.sh libA.so libB.so layer...
A=$1; B=$2; shift 2
for l in "$@"; do
  for lib in $A $B; do
    echo -n "$(basename $lib) "
    UNET_HIP_LIB=$PWD/$lib timeout -k 10 200 python -u tools/conv_microbench.py --iters 20 --only $l 2>&1 | grep "\->" | head -1 || exit 1
  done
done
