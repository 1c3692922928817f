#!/bin/bash
# Dev tool (round 5): same-box A/B of the pipeline object's stream / grouping knobs on the metric's line (fp32, default video).
W="--no-cpu-baseline --no-roofline --file-to-file 0 --supplementary none --steps 6 --warmup 1"
run() { echo -n "$1: "; env $2 python bench.py $W 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s', d['box'].get('sclk_mhz_mean_of_xcds',{}).get('mean'), 'MHz', d['box'].get('socket_power_w',{}).get('mean'), 'W')"; }
run "default (groups of 8 frames, 2 refinement lanes)" "X=1"
run "groups of 4, 4 lanes" "PREMVOS_REFINE_GROUP=4 PREMVOS_REFINE_LANES=4"
run "groups of 4, 2 lanes" "PREMVOS_REFINE_GROUP=4 PREMVOS_REFINE_LANES=2"
run "groups of 8, 1 lane" "PREMVOS_REFINE_LANES=1"
run "groups of 16, 1 lane" "PREMVOS_REFINE_GROUP=16 PREMVOS_REFINE_LANES=1"
run "serial stages" "PREMVOS_PIPELINE_SERIAL=1"
run "default again" "X=2"
