#!/bin/bash
# Dev experiment: A/B of two corr_tile.hip sources (stand-alone builds): time at the pyramid levels and bit-compare the outputs.
#   tools/dev/corr_ab.sh <old.hip> <new.hip>        (run from the repo root on the GPU box; the old file must sit next to common.h)
set -e
OUT=/tmp/corr_ab; mkdir -p $OUT
cp $1 premvos_amd/csrc/_corr_old_tmp.hip
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -DCORR_DBG_ENTRY -shared"
/opt/rocm/bin/hipcc $FLAGS premvos_amd/csrc/_corr_old_tmp.hip -o $OUT/old.so &
/opt/rocm/bin/hipcc $FLAGS $NEWFLAGS $2 -o $OUT/new.so &
wait
rm -f premvos_amd/csrc/_corr_old_tmp.hip
python - <<'PY'
import ctypes as C, torch
libs = {}
for n in ("old", "new"):
    lib = C.CDLL(f"/tmp/corr_ab/{n}.so")
    lib.corr_dbg.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 4 + [C.c_void_p]
    libs[n] = lib
for lvl, c, h, w in [(6, 196, 8, 14), (5, 128, 16, 28), (4, 96, 32, 56), (3, 64, 64, 112), (2, 32, 128, 224), (0, 20, 9, 33), (0, 40, 19, 70)]:
    B = 16
    f1 = torch.randn((B, h, w, c), device="cuda"); f2 = torch.randn((B, h, w, c), device="cuda")
    ps = 448 + 81 + c + 4
    outs = {}
    line = f"level {lvl} C={c:3d} {h:3d}x{w:3d}:"
    for n, lib in libs.items():
        out = torch.zeros((B, h, w, ps), device="cuda")
        def run():
            lib.corr_dbg(f1.data_ptr(), c, f2.data_ptr(), c, out.data_ptr() + 4 * 448, ps, B, h, w, c, torch.cuda.current_stream().cuda_stream)
        for _ in range(3): run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): run()
        b.record(); b.synchronize()
        byt = B * h * w * 4 * (2 * c + 81 + c)
        t = a.elapsed_time(b) / 20 * 1e3
        line += f"  {n} {t:7.1f} us {byt / t / 1e6:5.2f} TB/s"
        outs[n] = out
    print(line, " identical:", torch.equal(outs["old"], outs["new"]), flush=True)
PY
