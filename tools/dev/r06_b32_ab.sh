#!/bin/bash
# round 6: B = 32 frames / 320 crops per launch against the default B = 16 / 160 crops, same box, alternating
mkdir -p gpurun_out/r06
F="--steps 4 --warmup 1 --no-cpu-baseline --file-to-file 0 --supplementary none"
export PREMVOS_TUNE_CACHE=/tmp/premvos_b32_tune.json      # the B = 32 signatures the shipped table lacks are explored by wall clock once (PREMVOS_AUTOTUNE=full: a fair opponent for the tabled B = 16 shapes), reused by the second run
for rep in 1 2; do
  python bench.py $F > gpurun_out/r06/ab_b16_$rep.json 2> /dev/null
  PREMVOS_AUTOTUNE=full PREMVOS_REFINE_GROUP=16 python bench.py --batch 32 $F > gpurun_out/r06/ab_b32_$rep.json 2> /dev/null
done
PREMVOS_AUTOTUNE=full PREMVOS_REFINE_GROUP=16 python bench.py $F > gpurun_out/r06/ab_b16_g16.json 2> /dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06/ab_b*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        print(f.split('/')[-1], d['value'], 'fps  frac', r['frac'], 'family', r['igemm_family_frac'], 'canon us', r['canonical_layer']['us'], 'refine', r['per_stage_tflops']['refine'],
              'whole', r.get('whole_step_frac'), 'sclk', d['box'].get('sclk_mhz_mean_of_xcds', {}).get('mean'), 'by rule / explored', d['conv_configurations'].get('signatures_by_rule'), d['conv_configurations'].get('signatures_explored_by_time'), 'ms/step', d['ms_per_step'])
    except Exception as e:
        print(f, 'ERR', e)
PY
