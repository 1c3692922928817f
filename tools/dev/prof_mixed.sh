#!/bin/bash
# Dev: rocprofv3 kernel stats of the mixed-bf16x3 bench (serial stages) -> gpurun_out/<tag>_mixed_bf16x3_serial_kernel_stats.csv
TAG=${1:-r04}
OUT=$PWD/gpurun_out
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
export PREMVOS_PIPELINE_SERIAL=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_prof_mixed" -o bench -- \
  python "$REPO/bench.py" --precision mixed-bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --scaling weak --file-to-file 0 > "$OUT/${TAG}_prof_mixed.log" 2>&1
cp "$(find $OUT/${TAG}_prof_mixed -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_mixed_bf16x3_serial_kernel_stats.csv"
rm -rf "$OUT/${TAG}_prof_mixed"
