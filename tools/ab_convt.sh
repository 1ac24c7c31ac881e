#!/bin/bash
# same-box A/B of the ConvTranspose microbenchmark: base library vs current, interleaved twice
for rep in 1 2; do
  for lib in unet_amd/libunet_hip_base.so unet_amd/libunet_hip.so; do
    echo "== $(basename $lib)"
    UNET_HIP_LIB=$PWD/$lib timeout -k 10 200 python -u tools/convt_microbench.py 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
