#!/bin/bash
# Dev tool (round 5): effective shader clock + matrix-pipe busy of one GEMM launch configuration (counters-only passes).
#   tools/dev/r05_pmc_clock.sh TAG PATTERN cmd...
TAG=$1; PATTERN=$2; shift 2
REPO=$PWD; OUT=$PWD/gpurun_out/pmc_clock_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  (cd $REPO && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- "$@" > $OUT/p$i.log 2>&1) || tail -3 $OUT/p$i.log
done
cd $REPO
PATTERN=$PATTERN OUTD=$OUT python - <<'PY'
import csv, glob, collections, os
pat = os.environ["PATTERN"]; outd = os.environ["OUTD"]
acc = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in sorted(glob.glob(outd + '/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            dur[r['Counter_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(acc.items()):
    d = dur[k]
    print(f"  {k:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}  dur {sum(d)/len(d)/1e3:9.1f} us")
if 'GRBM_GUI_ACTIVE' in acc:
    g = sum(acc['GRBM_GUI_ACTIVE']) / len(acc['GRBM_GUI_ACTIVE']); d = sum(dur['GRBM_GUI_ACTIVE']) / len(dur['GRBM_GUI_ACTIVE'])
    print(f"  => effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) = {g / 8 / d:.3f} GHz ; raw/dur = {g / d:.3f}")
PY
