# A/B of the bf16x3 mode of the proposal net: HYBRID (3x3 on fp32 Winograd) x SPLIT (1x1 on the pure-bf16 staging kernel)
for cfg in "0 0" "1 0" "1 1"; do
set -- $cfg
PREMVOS_BF16X3_HYBRID=$1 PREMVOS_BF16X3_SPLIT_PROP=$2 python bench.py --precision mixed-bf16x3 --steps 4 --warmup 2 --no-cpu-baseline --scaling weak --file-to-file 0 2>&1 | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('hybrid=$1 split_prop=$2', d['value'], r.get('achieved'), r.get('per_stage_tflops'))"
done
