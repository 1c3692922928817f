#!/bin/bash
# what clock / power telemetry does the GPU box expose?
ls /sys/class/drm/ 2>&1 | head
for c in /sys/class/drm/card*/device; do
  echo "== $c"; ls $c | head -80
  cat $c/pp_dpm_sclk 2>&1 | head; cat $c/pp_dpm_mclk 2>&1 | head
  for h in $c/hwmon/hwmon*; do echo "-- $h"; ls $h; for f in $h/power1_average $h/power1_input $h/freq1_input $h/freq2_input $h/temp1_input $h/power1_cap; do echo "$f: $(cat $f 2>&1)"; done; done
done 2>&1 | head -150
which rocm-smi amd-smi
timeout 30 rocm-smi --showclocks --showpower --showtemp 2>&1 | head -40
timeout 30 amd-smi metric -g 0 --clock --power 2>&1 | head -60
python - <<'PY'
try:
    import amdsmi
    print("amdsmi importable", amdsmi.__file__)
except Exception as e:
    print("amdsmi import failed:", e)
PY
