#!/bin/bash
# Section 3 of tools/profile_round.sh on its own: the two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, serial stages)
# -> gpurun_out/<tag>_conv_hbm_traffic.json.  Run through gpurun from the repo root: tools/dev/traffic_only.sh r04b
set -u
TAG=${1:-dev}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
W="--scaling weak --file-to-file 0 --supplementary none"
REPO=$PWD
python bench.py --steps 2 --warmup 1 --no-cpu-baseline $W 2>/dev/null | grep '^{' | tail -1 > "$OUT/${TAG}_bench_weak.json"
STEPS=$(python -c "import json;print(json.load(open('$OUT/${TAG}_bench_weak.json'))['roofline']['launches_per_step'])")
cd /tmp && export TMPDIR=/tmp
export PREMVOS_PIPELINE_SERIAL=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_$c" -o pmc -- \
    python "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline $W > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
cd "$REPO"
python tools/pmc_traffic.py "$(find $OUT/${TAG}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
  "$(find $OUT/${TAG}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_conv_hbm_traffic.json" $STEPS
rm -rf "$OUT/${TAG}_pmc_FETCH_SIZE" "$OUT/${TAG}_pmc_WRITE_SIZE"
