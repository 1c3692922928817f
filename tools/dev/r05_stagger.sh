#!/bin/bash
# Dev tool (round 5): does starting refinement lane 1 a fraction of a layer late (so that its depthwise kernels meet lane 0's GEMMs
# instead of lane 0's depthwise kernels) help the metric's line?
# (needs the PREMVOS_LANE_STAGGER_CYCLES developer knob in premvos_amd/pipeline.py: torch.cuda._sleep on lane l before its refinement call; removed again)
W="--no-cpu-baseline --no-roofline --file-to-file 0 --supplementary none --steps 6 --warmup 1"
run() { echo -n "$1: "; env $2 python bench.py $W 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s', d['box'].get('sclk_mhz_mean_of_xcds',{}).get('mean'), 'MHz')"; }
run "no stagger" "X=1"
run "lane 1 starts 150 k cycles (~65 us) late" "PREMVOS_LANE_STAGGER_CYCLES=150000"
run "300 k (~130 us)" "PREMVOS_LANE_STAGGER_CYCLES=300000"
run "700 k (~300 us)" "PREMVOS_LANE_STAGGER_CYCLES=700000"
run "1100 k (~480 us)" "PREMVOS_LANE_STAGGER_CYCLES=1100000"
run "no stagger again" "X=2"
