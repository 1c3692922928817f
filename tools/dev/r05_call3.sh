#!/bin/bash
for t in tl tlnp tlep tlepnp; do
  TL_TAG=_$t PREMVOS_LIB_PATH=premvos_amd/csrc/libpremvos_hip_$t.so timeout 300 python tools/dev/r05_timeline.py 2>&1 | grep -v amdgpu.ids
done
