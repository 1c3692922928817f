#!/bin/bash
mkdir -p gpurun_out/r05
PREMVOS_LIB_PATH=premvos_amd/csrc/libpremvos_hip_tl.so timeout 300 python tools/dev/r05_timeline.py > gpurun_out/r05/timeline.txt 2>&1
timeout 120 python tools/dev/r05_smi_probe.py > gpurun_out/r05/smi_probe.txt 2>&1
tail -n 40 gpurun_out/r05/timeline.txt gpurun_out/r05/smi_probe.txt
