"""Dev tool (round 5): does a second stream fill the ragged ends of a GEMM launch?  N independent copies of one pointwise layer
on 1 / 2 / 3 streams: time per launch."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
lib = _lib.load()
for name, n, h, w, cin, cout in (("mid 728->728", 160, 25, 25, 728, 728), ("res conv3 256->1024", 16, 47, 84, 256, 1024), ("exit 1536->2048", 160, 25, 25, 1536, 2048)):
    descs = []
    for i in range(3):
        x = ops.NHWC(torch.randn((n, h, w, cin), device="cuda"), c=cin)
        pk = ops.pack_conv(torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5, torch.randn(cout) * 0.1)
        out = ops.NHWC.alloc(n, h, w, cout)
        descs.append((ops.conv_desc(x, pk, out, act=ops.ACT_RELU, tile_hint=(128 << 16) | 128, stage_k=16, split_k=-1), x, pk, out))
    streams = [torch.cuda.Stream() for _ in range(3)]
    res = []
    for ns in (1, 2, 3):
        REP = 12
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for s in streams[:ns]:
                s.wait_event(a)
            for r in range(REP):
                for i in range(ns):
                    lib.premvos_conv2d_f32(C.byref(descs[i][0]), C.c_void_p(streams[i].cuda_stream))
            for s in streams[:ns]:
                torch.cuda.current_stream().wait_stream(s)
            b.record(); b.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / (REP * ns))
        res.append(f"{ns} stream(s): {best:7.1f} us/launch {2.0 * n * h * w * cin * cout / best / 1e6:6.1f} TF/s")
    print(f"{name:22s} " + " | ".join(res), flush=True)
