#!/bin/bash
# The S8 kernel (tile 0) on a few layer shapes, once per library build named on the command line ("" = the shipped one):
#   tools/dev/s8_libs.sh "" _stag1 _stag2 ""
for lib in "$@"; do
  echo "== libpremvos_hip$lib.so"
  for only in "mid 728->728" "ksweep 256->728" "exit 1536->2048" "g2 conv1" "g2 conv3"; do
    PREMVOS_LIB_PATH=$PWD/premvos_amd/csrc/libpremvos_hip$lib.so S8_TILES=${S8_TILES:-0} S8_ONLY="$only" python tools/dev/s8_bench.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  %-42s f32out %8.1f us  s8out %8.1f us' % (d['layer'], d.get('t0_f32out_us', 0), d.get('t0_s8out_us', 0)))"
  done
done
