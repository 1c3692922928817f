#!/bin/bash
# Regenerates the judged evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01      -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
# 1. tuned-configuration cache + the default bench line   2. rocprofv3 kernel stats, serial and concurrent stages
# 3. PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) -> HBM bytes per conv launch
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
# (round 3: the conv configurations come from the shipped table premvos_amd/tune_gfx950.json -- tools/make_tune_table.py -- so
#  there is no per-run tune cache to fill any more; the profiled runs use one pipeline step per bench step: --scaling weak)
cp premvos_amd/tune_gfx950.json "$OUT/${TAG}_tune_choices.json"
W="--scaling weak --file-to-file 0 --supplementary none"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for mode in serial concurrent; do
  [ $mode = serial ] && export PREMVOS_PIPELINE_SERIAL=1 || unset PREMVOS_PIPELINE_SERIAL
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_prof_$mode" -o bench -- \
    python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline $W > "$OUT/${TAG}_prof_$mode.log" 2>&1
  grep '^{' "$OUT/${TAG}_prof_$mode.log" | tail -1 > "$OUT/${TAG}_bench_under_rocprof_$mode.json"
done
export PREMVOS_PIPELINE_SERIAL=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/${TAG}_pmc_$c" -o pmc -- \
    python "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline $W > "$OUT/${TAG}_pmc_$c.log" 2>&1
done
unset PREMVOS_PIPELINE_SERIAL
cd "$REPO"
STEPS=$(python -c "import json;print(json.load(open('$OUT/${TAG}_bench_under_rocprof_serial.json'))['roofline']['launches_per_step'])")
python tools/pmc_traffic.py "$(find $OUT/${TAG}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
  "$(find $OUT/${TAG}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" "$OUT/${TAG}_conv_hbm_traffic.json" $STEPS \
  > /dev/null
cp "$OUT/${TAG}_conv_hbm_traffic.json" profiles/${TAG}_conv_hbm_traffic.json      # bench.py reads the traffic figure from here
# 4. the cost-volume kernel on its own (rocprofv3 kernel stats of tools/time_corr.py: the 5 pyramid levels of a 16-pair step)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_prof_corr" -o corr -- \
   python "$REPO/tools/time_corr.py" 16 > "$OUT/${TAG}_time_corr.log" 2>&1)
cp "$(find $OUT/${TAG}_prof_corr -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_corr_kernel_stats.csv"; rm -rf "$OUT/${TAG}_prof_corr"
# 5. two ranks through the self-launching path (gloo: both ranks share this box's one GPU; RCCL needs a device per rank)
PREMVOS_BENCH_BACKEND=gloo python bench.py --gpus 2 --batch 4 --frames 30 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline \
   --file-to-file 0 > "$OUT/${TAG}_bench_2ranks_gloo.log" 2>&1           # strong scaling: 30 frame pairs = 8 chunks of 4 (last ragged) over 2 ranks
grep '^{' "$OUT/${TAG}_bench_2ranks_gloo.log" | tail -1 > "$OUT/${TAG}_bench_2ranks_gloo.json"
# 6. the optional bf16-MFMA modes (configs[2] / [4] name bf16; never the headline): one line each, priced against the bf16 peak
for prec in mixed-bf16x3 mixed-bf16; do
  python bench.py --precision $prec --steps 5 --warmup 2 --no-cpu-baseline $W > "$OUT/${TAG}_bench_$prec.log" 2>&1
  grep '^{' "$OUT/${TAG}_bench_$prec.log" | tail -1 > "$OUT/${TAG}_bench_$prec.json"
done
python tools/layer_table.py > "$OUT/${TAG}_layer_table.txt" 2>&1
# 7. round 4: the bf16x3 mode on the resident S8 layout -- per-layer table, rocprofv3 kernel stats (serial stages), the kernel per
#    layer shape and tile against the fp32 kernels, configs[4]'s 1080p shape in both arithmetics; the merge-shaped serial loop
LT_PRECISION=bf16x3 python tools/layer_table.py > "$OUT/${TAG}_layer_table_bf16x3.txt" 2>&1
tools/dev/prof_mixed.sh "$TAG"
python tools/dev/s8_bench.py > "$OUT/${TAG}_s8_bench.jsonl" 2> /dev/null
for prec in fp32 mixed-bf16x3; do
  python bench.py --frame 1080p --precision $prec --steps 5 --warmup 2 --file-to-file 0 > "$OUT/${TAG}_bench_1080p_$prec.log" 2>&1
  grep '^{' "$OUT/${TAG}_bench_1080p_$prec.log" | tail -1 > "$OUT/${TAG}_bench_1080p_$prec.json"
done
python tools/time_merge_loop.py --out "$OUT/${TAG}_merge_loop.json" > /dev/null 2>&1
# 8. round 6: the merge rank of a gathered 8-rank job on this one GPU (recorded buffers of 7 ranks replayed beside the real driver), in
#    the shipped form and in round 5's; where a rank's cold start goes
python tools/time_merge_ingest.py --frames 128 --legacy --out "$OUT/${TAG}_merge_ingest.json" > "$OUT/${TAG}_merge_ingest.log" 2>&1
python tools/dev/cold_start_profile.py > "$OUT/${TAG}_cold_start_profile.txt" 2>&1
python bench.py > "$OUT/${TAG}_bench_fp32.log" 2>&1
grep '^{' "$OUT/${TAG}_bench_fp32.log" | tail -1 > "$OUT/${TAG}_bench_fp32.json"
# keep only the small summaries (the traces are tens of MB)
for mode in serial concurrent; do
  cp "$(find $OUT/${TAG}_prof_$mode -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_fp32_${mode}_kernel_stats.csv"
  rm -rf "$OUT/${TAG}_prof_$mode"
done
rm -rf "$OUT/${TAG}_pmc_FETCH_SIZE" "$OUT/${TAG}_pmc_WRITE_SIZE"
cat "$OUT/${TAG}_bench_fp32.json"
