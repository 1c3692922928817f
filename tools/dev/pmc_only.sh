#!/bin/bash
# Dev tool: only the two PMC passes of tools/profile_round.sh (HBM bytes per conv launch) -> gpurun_out/<tag>_conv_hbm_traffic.json
set -u
TAG=${1:-r03}; OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p "$OUT"
W="--scaling weak --file-to-file 0"
cd /tmp && export TMPDIR=/tmp
export PREMVOS_PIPELINE_SERIAL=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_$c" -o pmc -- \
    python "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline $W > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
cd "$REPO"
STEPS=$(python -c "import json;print(json.load(open('profiles/${TAG}_bench_fp32.json'))['roofline']['launches_per_step'])")
python tools/pmc_traffic.py "$(find $OUT/${TAG}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
  "$(find $OUT/${TAG}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_conv_hbm_traffic.json" $STEPS | tail -5
rm -rf "$OUT/${TAG}_pmc_FETCH_SIZE" "$OUT/${TAG}_pmc_WRITE_SIZE"
