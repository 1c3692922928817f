#!/bin/bash
# Dev tool: per-kernel time of the three F(4x4,3x3) launches on the pipeline's K-rich 3x3 layers (rocprofv3 kernel stats).
#   tools/dev/wino4_steps.sh     (on the GPU box, from the repo root)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "16 46 83 1024 1024" "16 46 83 256 256" "1600 7 7 512 512" "16 64 112 277 128"; do
  echo "== n h w cin cout = $cfg"
  rm -rf /tmp/prof_w4
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w4 -o w4 -- python $REPO/tools/dev/one_wino.py $cfg 4 ${SK:-0} 2>&1 | grep "TF/s"
  python - <<PY
import csv, subprocess
f = subprocess.run("find /tmp/prof_w4 -name '*kernel_stats.csv' | head -1", shell=True, capture_output=True, text=True).stdout.strip()
for r in csv.DictReader(open(f)):
    if "wino4" in r["Name"]:
        print("    %-62s calls %3s  avg %8.1f us" % (r["Name"].split("(")[0][-62:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
