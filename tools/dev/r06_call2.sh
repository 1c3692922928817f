#!/bin/bash
# round 6, GPU call 2: full GPU suite (device handling changed everywhere), merge ingest with the C frame writer + object-like masks, cold-start profile
mkdir -p gpurun_out/r06
timeout 900 python tools/time_merge_ingest.py --frames 64 --legacy --out gpurun_out/r06/merge_ingest2.json > gpurun_out/r06/merge_ingest2.log 2>&1
echo "ingest rc=$?"; tail -n 3 gpurun_out/r06/merge_ingest2.log | cut -c1-3000
timeout 600 python tools/dev/cold_start_profile.py > gpurun_out/r06/cold_start_profile.txt 2>&1
echo "cold rc=$?"; head -n 12 gpurun_out/r06/cold_start_profile.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06/tests2.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r06/tests2.txt
tail -n 15 gpurun_out/r06/tests2.txt
