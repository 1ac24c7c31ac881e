#!/bin/bash
# same-box A/B of the conv microbenchmark: base library vs current, interleaved
for l in "$@"; do
  for lib in unet_amd/libunet_hip_base.so unet_amd/libunet_hip.so; do
    echo -n "$(basename $lib) "
    UNET_HIP_LIB=$PWD/$lib timeout -k 10 200 python -u tools/conv_microbench.py --iters 20 --only $l 2>&1 | grep "\->" || exit 1
  done
done
