set -u
TAG=r02; OUT=$PWD/gpurun_out; REPO=$PWD
export PREMVOS_TUNE_CACHE=$OUT/${TAG}_tune_choices.json
rm -f "$PREMVOS_TUNE_CACHE"
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
export PREMVOS_PIPELINE_SERIAL=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_$c" -o pmc -- \
    python "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
unset PREMVOS_PIPELINE_SERIAL
cd "$REPO"
python tools/pmc_traffic.py "$(find $OUT/${TAG}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
  "$(find $OUT/${TAG}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_conv_hbm_traffic.json" 435
rm -rf "$OUT/${TAG}_pmc_FETCH_SIZE" "$OUT/${TAG}_pmc_WRITE_SIZE"
