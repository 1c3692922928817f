#!/bin/bash
# Dev tool: PMC passes (counters only + kernel trace, one group per pass) over any command; prints the mean counter value
# per dispatch of the kernels whose name contains PATTERN, grouped by grid size.
#   tools/pmc_kernel.sh corr81_tile python tools/time_corr.py 16
PATTERN=$1; shift
GROUPS_FILTER=${PMC_GROUPS:-all}      # e.g. PMC_GROUPS="1 6 7 8" runs only those counter groups
REPO=$PWD; OUT=$PWD/gpurun_out/pmc_kernel; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  if [ "$GROUPS_FILTER" != all ] && ! echo " $GROUPS_FILTER " | grep -q " $i "; then continue; fi
  (cd $REPO && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- "$@" > $OUT/p$i.log 2>&1)
done
cd $REPO
PATTERN=$PATTERN python - <<'PY'
import csv, glob, collections, os
pat = os.environ["PATTERN"]
acc = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/pmc_kernel/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            key = (r.get('Grid_Size', '?'), r['Counter_Name'])
            key = (key[0].rjust(10), key[1])
            acc[key].append(float(r['Counter_Value']))
            if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
                dur[r.get('Grid_Size', '?')].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for (g, k), v in sorted(acc.items()):
    print(f"grid {g:>10s} {k:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
for g, v in dur.items():
    print(f"grid {g}: mean duration under PMC {sum(v)/len(v)/1e3:.1f} us")
PY
