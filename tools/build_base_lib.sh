#!/bin/bash
# A/B support: builds unet_amd/libunet_hip_base.so from the current sources with the files named on the command line
# taken from a git revision (default HEAD).  Use with UNET_HIP_LIB=unet_amd/libunet_hip_base.so on the same GPU box.
#   tools/build_base_lib.sh [rev] file...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:-HEAD}; shift || true
T=/tmp/base_csrc
rm -rf $T; mkdir -p $T
cp $ROOT/unet_amd/csrc/*.hip $ROOT/unet_amd/csrc/*.h $ROOT/unet_amd/csrc/*.cpp $ROOT/unet_amd/csrc/Makefile $T/
for f in "$@"; do git -C $ROOT show $REV:unet_amd/csrc/$f > $T/$f; done
make -C $T -j8 LIB=$ROOT/unet_amd/libunet_hip_base.so > /tmp/base_build.log 2>&1 || { tail -20 /tmp/base_build.log; exit 1; }
ls -la $ROOT/unet_amd/libunet_hip_base.so
