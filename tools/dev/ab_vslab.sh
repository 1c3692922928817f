#!/bin/bash
# Same-box A/B of the kept Winograd slabs (PREMVOS_VSLAB=0 / 1) on the default bench line; run through gpurun.
mkdir -p gpurun_out
for v in 0 1 0 1; do
  PREMVOS_VSLAB=$v timeout 300 python bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/ab_vslab_$v.json
  python - "$v" <<'P'
import json, sys
d = json.load(open(f"gpurun_out/ab_vslab_{sys.argv[1]}.json"))
print(f"PREMVOS_VSLAB={sys.argv[1]}  {d['value']:.2f} frames/s  {d['ms_per_step']:.1f} ms/step  roofline frac {d['roofline'].get('frac')}")
P
done
