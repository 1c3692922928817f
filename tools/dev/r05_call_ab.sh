#!/bin/bash
# usage: r05_call_ab.sh tagA tagB ...  (tag "base" = the committed library)
for rep in 1 2; do
for t in "$@"; do
  if [ $t = base ]; then unset PREMVOS_LIB_PATH; else export PREMVOS_LIB_PATH=premvos_amd/csrc/libpremvos_hip_$t.so; fi
  echo "=== $t (rep $rep)"; timeout 300 python tools/dev/r05_ab.py 2>&1 | grep -v amdgpu.ids
done
done
