#!/bin/bash
# Dev experiment: which phase of corr81_tile_kernel bounds it?  Builds stand-alone variants with phases compiled out
# (they compute garbage) and times them at the 128x224 / 64x112 levels.  Run from the repo root on the GPU box.
set -e
OUT=/tmp/corr_variants; mkdir -p $OUT
for v in full:"" nocompute:"-DCORR_DBG_NO_COMPUTE" nooutput:"-DCORR_DBG_NO_OUTPUT" noload:"-DCORR_DBG_NO_LOAD" \
         computeonly:"-DCORR_DBG_NO_OUTPUT -DCORR_DBG_NO_LOAD" loadonly:"-DCORR_DBG_NO_OUTPUT -DCORR_DBG_NO_COMPUTE"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -DCORR_DBG_ENTRY $flags -shared \
     premvos_amd/csrc/corr_tile.hip -o $OUT/corr_$name.so &
done
wait
python - <<'PY'
import ctypes as C, torch, glob
for lvl, c, h, w in [(3, 64, 64, 112), (2, 32, 128, 224)]:
    B = 16
    f1 = torch.randn((B, h, w, c), device="cuda"); f2 = torch.randn((B, h, w, c), device="cuda")
    ps = 448 + 81 + c + 4
    out = torch.zeros((B, h, w, ps), device="cuda")
    for so in sorted(glob.glob("/tmp/corr_variants/corr_*.so")):
        lib = C.CDLL(so)
        lib.corr_dbg.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 4 + [C.c_void_p]
        def run():
            lib.corr_dbg(f1.data_ptr(), c, f2.data_ptr(), c, out.data_ptr() + 4 * 448, ps, B, h, w, c, torch.cuda.current_stream().cuda_stream)
        for _ in range(3): run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): run()
        b.record(); b.synchronize()
        print(f"level {lvl}  {so.split('corr_')[-1][:-3]:12s} {a.elapsed_time(b) / 20 * 1e3:8.1f} us", flush=True)
PY
