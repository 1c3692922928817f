#!/bin/bash
# round 5, GPU call 1: telemetry probe, tile A/B, clock / pipe-busy PMC
mkdir -p gpurun_out/r05
bash tools/dev/r05_sysfs_probe.sh > gpurun_out/r05/sysfs_probe.txt 2>&1
timeout 600 python tools/dev/r05_tiles.py > gpurun_out/r05/tiles.txt 2>&1
timeout 300 bash tools/dev/r05_pmc_clock.sh calib mfma_f32_calibrate python tools/dev/one_calib.py > gpurun_out/r05/pmc_calib.txt 2>&1
timeout 300 bash tools/dev/r05_pmc_clock.sh k3072_128 conv_igemm python tools/one_conv.py 96 32 32 3072 768 1 128 128 16 > gpurun_out/r05/pmc_k3072_128.txt 2>&1
timeout 300 bash tools/dev/r05_pmc_clock.sh k3072_128x256 conv_igemm python tools/one_conv.py 96 32 32 3072 768 1 128 256 16 > gpurun_out/r05/pmc_k3072_128x256.txt 2>&1
timeout 300 bash tools/dev/r05_pmc_clock.sh mid728 conv_igemm python tools/one_conv.py 160 25 25 728 728 1 128 128 16 > gpurun_out/r05/pmc_mid728.txt 2>&1
timeout 200 python tools/dev/power_data.py > gpurun_out/r05/power_data.txt 2>&1
tail -n 30 gpurun_out/r05/tiles.txt gpurun_out/r05/pmc_*.txt
