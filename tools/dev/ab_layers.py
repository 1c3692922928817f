"""Dev tool: the big pointwise layers of the refinement net (+ two ResNet shapes) on fixed tile configurations; run once per
library build (PREMVOS_LIB_PATH=...) on the same box for A/B comparisons of kernel changes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
import ctypes as C
lib, st = _lib.load(), _lib.current_stream()
LAYERS = [("mid 728->728 25x25x160", 160, 25, 25, 728, 728, 1), ("mid 728->728 49x49x160", 160, 49, 49, 728, 728, 1),
          ("exit 728->1024", 160, 25, 25, 728, 1024, 1), ("exit 1024->1536", 160, 25, 25, 1024, 1536, 1),
          ("exit 1536->2048", 160, 25, 25, 1536, 2048, 1), ("entry 256->728 49x49", 160, 49, 49, 256, 728, 1),
          ("res conv4 1024->256", 16, 47, 84, 1024, 256, 1), ("ideal 1024->1024 128x128x8", 8, 128, 128, 1024, 1024, 1)]
for name, n, h, w, cin, cout, k in LAYERS:
    x = ops.NHWC(torch.randn((n, h, w, (cin + 3) // 4 * 4), device="cuda"), c=cin)
    out = ops.NHWC.alloc(n, h, w, cout)
    pk = ops.pack_conv(torch.randn((cout, cin, k, k)) * (2.0 / (k * k * cin)) ** 0.5, torch.zeros(cout))
    res = []
    for bm, bn, sk in ((128, 128, 16), (128, 128, 32), (256, 128, 16)):
        d = ops.conv_desc(x, pk, out, pad=(k // 2, k // 2), act=ops.ACT_RELU, tile_hint=(bm << 16) | bn, stage_k=sk, split_k=-1)
        for _ in range(3):
            _lib.check(lib.premvos_conv2d_f32(C.byref(d), st))
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                lib.premvos_conv2d_f32(C.byref(d), st)
            b.record(); b.synchronize()
            best = min(best, a.elapsed_time(b) * 100)
        res.append(f"{bm}x{bn}/{sk}: {best:7.1f} us {2.0 * n * h * w * k * k * cin * cout / best / 1e6:6.1f} TF/s")
    print(f"{name:28s}", " | ".join(res), flush=True)
