#!/bin/bash
# same-box timing of the diagnostic variants of the conv kernel (wrong results by construction: timing only)
#   tools/ab_variants.sh layer...   with unet_amd/libunet_hip_x_<variant>.so built by make EXTRA=-DCV2_X_<VARIANT>
for l in "$@"; do
  for lib in unet_amd/libunet_hip.so unet_amd/libunet_hip_x_*.so; do
    echo -n "$(basename $lib) "
    UNET_HIP_LIB=$PWD/$lib timeout -k 10 200 python -u tools/conv_microbench.py --iters 20 --only $l 2>&1 | grep "\->" | head -1 || exit 1
  done
done
