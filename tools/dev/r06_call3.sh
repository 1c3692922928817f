#!/bin/bash
# round 6, GPU call 3: ingest on a 128-frame clip (steady-state figure), B = 32 A/B, tests of everything that packs weights
mkdir -p gpurun_out/r06
timeout 900 python tools/time_merge_ingest.py --frames 128 --out gpurun_out/r06/merge_ingest3.json > gpurun_out/r06/merge_ingest3.log 2>&1
echo "ingest rc=$?"; tail -n 2 gpurun_out/r06/merge_ingest3.log | cut -c1-2500
timeout 1500 python -m pytest tests/test_gpu_bench_object.py tests/test_gpu_pwc.py tests/test_gpu_proposal.py tests/test_gpu_refinement.py tests/test_gpu_conv_wino.py tests/test_gpu_conv_s8.py tests/test_gpu_conv_pwdma.py tests/test_gpu_precision_modes.py tests/test_gpu_reid.py tests/test_gpu_bench_contract.py -x -q > gpurun_out/r06/tests3.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r06/tests3.txt; tail -n 6 gpurun_out/r06/tests3.txt
timeout 1500 bash tools/dev/r06_b32_ab.sh 2>&1 | tail -n 12
