#!/bin/bash
# Dev tool: PMC passes over one conv shape (tools/conv_one.py <shape>), mean counter value per conv dispatch.
#   TILE=128x128 STAGE=16 tools/pmc_conv.sh ideal
SHAPE=${1:-ideal}
REPO=$PWD; OUT=$PWD/gpurun_out/pmc_conv; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $REPO/tools/conv_one.py $SHAPE > $OUT/p$i.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list); dur = []
for f in sorted(glob.glob('gpurun_out/pmc_conv/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] in ('GRBM_GUI_ACTIVE',): dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in acc.items():
    print(f"{k:32s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
if dur: print("mean duration ns (under PMC):", sum(dur)/len(dur))
PY
