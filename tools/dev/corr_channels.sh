set -e
OUT=/tmp/corr_variants; mkdir -p $OUT
for v in full:"" loadonly:"-DCORR_DBG_NO_OUTPUT -DCORR_DBG_NO_COMPUTE"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -fno-slp-vectorize -DCORR_DBG_ENTRY $flags -shared \
     premvos_amd/csrc/corr_tile.hip -o $OUT/corr_$name.so &
done
wait
python - <<'PY'
import ctypes as C, torch, glob
for c in (16, 32, 64):
    B, h, w = 16, 128, 224
    f1 = torch.randn((B, h, w, c), device="cuda"); f2 = torch.randn((B, h, w, c), device="cuda")
    ps = 448 + 81 + c + 4
    out = torch.zeros((B, h, w, ps), device="cuda")
    for name in ("loadonly", "full"):
        lib = C.CDLL(f"/tmp/corr_variants/corr_{name}.so")
        lib.corr_dbg.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_int] * 4 + [C.c_void_p]
        def run():
            lib.corr_dbg(f1.data_ptr(), c, f2.data_ptr(), c, out.data_ptr() + 4 * 448, ps, B, h, w, c, torch.cuda.current_stream().cuda_stream)
        for _ in range(3): run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): run()
        b.record(); b.synchronize()
        t = a.elapsed_time(b) / 20 * 1e3
        print(f"C={c:3d} 128x224 {name:9s} {t:8.1f} us   input {2*B*h*w*c*4/1e6:6.1f} MB -> {2*B*h*w*c*4/t/1e6:5.2f} TB/s of unique input", flush=True)
PY
