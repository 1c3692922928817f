#!/usr/bin/env python
"""Dev tool: per-kernel register / scratch / occupancy table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import build as b
cmd = [b._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", *b._file_flags(src),
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for ln in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)", ln)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split()[0]] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(premvos_conv_desc.*", "", name)
    if flt in name:
        g = lambda k: str(r.get(k))
        print(f"{name[:90]:90s} vgpr {g('VGPRs'):>4} agpr {g('AGPRs'):>4} sgpr {g('SGPRs'):>4} scratch {g('ScratchSize'):>4} occ {g('Occupancy')}")
