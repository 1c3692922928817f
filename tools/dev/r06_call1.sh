#!/bin/bash
# round 6, GPU call 1: the changed paths' tests, the merge-ingest measurement (round-6 and round-5 form), the default bench line
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_merge.py tests/test_gpu_arena.py tests/test_gpu_plumbing.py tests/test_gpu_refinement.py tests/test_gpu_proposal.py -x -q > gpurun_out/r06/tests1.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r06/tests1.txt
tail -n 15 gpurun_out/r06/tests1.txt
timeout 900 python tools/time_merge_ingest.py --frames 64 --legacy --out gpurun_out/r06/merge_ingest.json > gpurun_out/r06/merge_ingest.log 2>&1
echo "ingest rc=$?"; tail -n 5 gpurun_out/r06/merge_ingest.log
timeout 900 python bench.py > gpurun_out/r06/bench1.json 2> gpurun_out/r06/bench1.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/r06/bench1.json; tail -n 5 gpurun_out/r06/bench1.err
