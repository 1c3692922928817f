"""Dev tool (round 5): per-workgroup phase stamps of the implicit GEMM (library built with -DPV_DBG_TIMELINE):
   PREMVOS_LIB_PATH=premvos_amd/csrc/libpremvos_hip_tl.so python tools/dev/r05_timeline.py
Saves gpurun_out/r05/timeline_<name>.npy: [wg, 8] = t_entry, t_prologue_done, t_loop_done, t_epilogue_done (shader cycles),
wall_entry (100 MHz), HW_ID, XCC_ID, wall_end."""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
lib, st = _lib.load(), _lib.current_stream()
raw = C.CDLL(os.environ["PREMVOS_LIB_PATH"])
raw.premvos_dbg_timeline.argtypes = [C.c_void_p, C.c_long]
os.makedirs("gpurun_out/r05", exist_ok=True)
TAG = os.environ.get("TL_TAG", "")
SHAPES = [("mid728", 160, 25, 25, 728, 728, False), ("k3072", 96, 32, 32, 3072, 768, False), ("k256", 96, 32, 32, 256, 768, False),
          ("res_conv3", 16, 47, 84, 256, 1024, True)]
for name, n, h, w, cin, cout, res in SHAPES:
    x = ops.NHWC(torch.randn((n, h, w, (cin + 3) // 4 * 4), device="cuda"), c=cin)
    pk = ops.pack_conv(torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5, torch.randn(cout) * 0.1)
    out = ops.NHWC.alloc(n, h, w, cout)
    r = ops.NHWC(torch.randn((n, h, w, cout), device="cuda"), c=cout) if res else None
    d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU, tile_hint=(128 << 16) | 128, stage_k=16, split_k=-1, **({"res": r} if res else {}))
    import time
    t_end = time.perf_counter() + 1.0                      # a second of back-to-back launches: the clocks have ramped, the chip is warm
    while time.perf_counter() < t_end:
        for _ in range(20):
            _lib.check(lib.premvos_conv2d_f32(C.byref(d), st))
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        lib.premvos_conv2d_f32(C.byref(d), st)
    a.record(); lib.premvos_conv2d_f32(C.byref(d), st); b.record(); b.synchronize()
    nwg = ((n * h * w + 127) // 128) * ((cout + 127) // 128)
    buf = np.zeros((nwg, 8), dtype=np.uint64)
    rc = raw.premvos_dbg_timeline(buf.ctypes.data, buf.nbytes)
    np.save(f"gpurun_out/r05/timeline_{name}{TAG}.npy", buf)
    print(name + TAG, "rc", rc, "nwg", nwg, "launch us", a.elapsed_time(b) * 1e3, flush=True)
