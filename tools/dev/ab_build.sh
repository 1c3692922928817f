#!/bin/bash
# Dev tool: build a second copy of the library with extra -D flags for same-box A/B runs:
#   tools/dev/ab_build.sh noskip -DPV_DBG_NOSKIP   ->  premvos_amd/csrc/libpremvos_hip_noskip.so
#   PREMVOS_LIB_PATH=premvos_amd/csrc/libpremvos_hip_noskip.so python tools/one_conv.py ...
set -e
cd "$(dirname "$0")/../.."
TAG=$1; shift
OUT=premvos_amd/csrc/build_$TAG
mkdir -p $OUT
for f in premvos_amd/csrc/*.hip; do
  extra=$(head -5 $f | grep -o "hipcc-flags:.*" | sed 's/hipcc-flags://')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result $extra "$@" -c $f -o $OUT/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o premvos_amd/csrc/libpremvos_hip_$TAG.so $OUT/*.o
echo built premvos_amd/csrc/libpremvos_hip_$TAG.so
