"""Dev tool (round 5): per-tile stamps of the persistent pointwise kernel (library built with -DPV_DBG_TIMELINE)."""
import os, sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
lib, st = _lib.load(), _lib.current_stream()
raw = C.CDLL(os.environ["PREMVOS_LIB_PATH"])
raw.premvos_dbg_timeline_pw.argtypes = [C.c_void_p, C.c_long]
os.makedirs("gpurun_out/r05", exist_ok=True)
TAG = os.environ.get("TL_TAG", "")
for name, n, h, w, cin, cout in (("mid728", 160, 25, 25, 728, 728), ("k1024", 96, 32, 32, 1024, 768)):
    x = ops.NHWC(torch.randn((n, h, w, cin), device="cuda"), c=cin)
    pk = ops.pack_conv(torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5, torch.randn(cout) * 0.1)
    out = ops.NHWC.alloc(n, h, w, cout)
    d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU, tile_hint=6, stage_k=16, split_k=-1)
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        for _ in range(20):
            _lib.check(lib.premvos_conv2d_f32(C.byref(d), st))
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        lib.premvos_conv2d_f32(C.byref(d), st)
    a.record(); lib.premvos_conv2d_f32(C.byref(d), st); b.record(); b.synchronize()
    buf = np.zeros((768, 16, 4), dtype=np.uint64)
    rc = raw.premvos_dbg_timeline_pw(buf.ctypes.data, buf.nbytes)
    np.save(f"gpurun_out/r05/timeline_pw_{name}{TAG}.npy", buf)
    print(name + TAG, "rc", rc, "launch us", a.elapsed_time(b) * 1e3, flush=True)
