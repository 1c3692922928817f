#!/bin/bash
# round 6: the cost-volume kernel's variants (csrc/corr_tile.hip: tile, chunk depth, occupancy, prefetch), one process each
mkdir -p gpurun_out/r06
for v in ${VARIANTS:-0 1 2 3 5 6 7 8 9 10 0}; do
  echo "== PREMVOS_CORR_VARIANT=$v"
  PREMVOS_CORR_VARIANT=$v timeout 300 python tools/time_corr.py 16 2>&1 | grep -E "^level" | sed -e 's/| warp+corr fused.*//'
done
