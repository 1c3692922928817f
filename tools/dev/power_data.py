"""Dev tool: is the implicit GEMM limited by the chip's power management?  Same launches, operands all zero / small-integer / random
(switching activity in the matrix pipe and the LDS differs, the instruction stream does not): TF/s per data class."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
import ctypes as C
lib, st = _lib.load(), _lib.current_stream()
for name, n, h, w, cin, cout in (("mid 728->728 M=100k", 160, 25, 25, 728, 728), ("ideal 1024->1024 M=131k", 8, 128, 128, 1024, 1024)):
    for data in ("random", "zeros", "ones", "random_relu", "random"):
        x = ops.NHWC.alloc(n, h, w, cin)
        wt = torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5
        if data == "random":
            x.buf.normal_()
        elif data == "random_relu":
            x.buf.normal_().clamp_(min=0)
        elif data == "ones":
            x.buf.fill_(1.0); wt.fill_(1.0)
        else:
            wt.zero_()
        out = ops.NHWC.alloc(n, h, w, cout)
        pk = ops.pack_conv(wt, torch.zeros(cout))
        d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU, tile_hint=(128 << 16) | 128, stage_k=16, split_k=-1)
        for _ in range(5):
            _lib.check(lib.premvos_conv2d_f32(C.byref(d), st))
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                lib.premvos_conv2d_f32(C.byref(d), st)
            b.record(); b.synchronize()
            best = min(best, a.elapsed_time(b) * 50)
        print(f"{name:26s} {data:12s} {best:8.1f} us {2.0 * n * h * w * cin * cout / best / 1e6:6.1f} TF/s", flush=True)
