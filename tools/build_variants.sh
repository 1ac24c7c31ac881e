#!/bin/bash
# Timing-only variants of the working tree's library for same-box A/B runs: bash tools/build_variants.sh name1:-DFLAG1 name2:"-DA -DB" ...
# builds _ab_<name>/unet_amd/libunet_hip.so (git-ignored, travels with gpurun); select with UNET_HIP_LIB.  Remove _ab_*/ when done.
set -e
cd "$(git rev-parse --show-toplevel)"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  rm -rf _ab_$name && mkdir -p _ab_$name/unet_amd
  cp -r unet_amd/csrc _ab_$name/unet_amd/csrc
  rm -rf _ab_$name/unet_amd/csrc/build
  make -C _ab_$name/unet_amd/csrc -j8 EXTRA="$flags" > /dev/null
  rm -rf _ab_$name/unet_amd/csrc
  echo "built _ab_$name/unet_amd/libunet_hip.so with $flags"
done
