#!/bin/bash
# Where the fixed cost of an S8 launch goes: the shipped kernel against builds without global stores / without the epilogue
# (tools/dev/ab_build.sh nostore -DPV_DBG_S8_NOSTORE; ... noepi -DPV_DBG_S8_NOEPI), tile 0, the middle flow's layer and the K sweep.
for lib in "" _nostore _noepi; do
  echo "== libpremvos_hip$lib.so"
  for only in "mid 728->728" "ksweep 256->728" "ksweep 1456->728" "exit 1536->2048" "g2 conv1"; do
    PREMVOS_LIB_PATH=$PWD/premvos_amd/csrc/libpremvos_hip$lib.so S8_TILES=${S8_TILES:-0} S8_ONLY="$only" python tools/dev/s8_bench.py 2>/dev/null | tail -2
  done
done
