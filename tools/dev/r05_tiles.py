"""Dev tool (round 5): wave-tile shape A/B of the fp32 implicit GEMM on padding-free 1x1 layers + the 728-wide middle-flow layer.
Every tile multiplies in the same k order: outputs must be bit-identical to the 128x128 tile's (checked)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
import ctypes as C
lib, st = _lib.load(), _lib.current_stream()
TILES = [("128x128", (128 << 16) | 128), ("pwdma", 6), ("256x128w8", (256 << 16) | 128), ("128x256", (128 << 16) | 256),
         ("256x128w4", (256 << 16) | 129), ("256x256w8", (256 << 16) | 256)]
SHAPES = [("sweep K=256", 96, 32, 32, 256, 768), ("sweep K=768", 96, 32, 32, 768, 768), ("sweep K=1024", 96, 32, 32, 1024, 768),
          ("sweep K=3072", 96, 32, 32, 3072, 768), ("mid 728->728", 160, 25, 25, 728, 728), ("exit 1536->2048", 160, 25, 25, 1536, 2048),
          ("res 1024->256", 16, 47, 84, 1024, 256), ("res 256->1024", 16, 47, 84, 256, 1024), ("entry 256->256", 160, 97, 97, 256, 256),
          ("entry 256->728", 160, 49, 49, 256, 728), ("exit 1024->1536", 160, 25, 25, 1024, 1536)]
if len(sys.argv) > 1 and sys.argv[1] == "dma":
    TILES = TILES[:2]
for name, n, h, w, cin, cout in SHAPES:
    x = ops.NHWC(torch.randn((n, h, w, (cin + 3) // 4 * 4), device="cuda"), c=cin)
    pk = ops.pack_conv(torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5, torch.randn(cout) * 0.1)
    ref = None
    res = []
    for tname, hint in TILES:
        out = ops.NHWC.alloc(n, h, w, cout)
        d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU, tile_hint=hint, stage_k=16, split_k=-1)
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
        o = out.buf.clone()
        if ref is None:
            ref = o
        same = bool(torch.equal(o, ref))
        res.append(f"{tname}: {best:7.1f} us {2.0 * n * h * w * cin * cout / best / 1e6:6.1f}{'' if same else ' MISMATCH'}")
    print(f"{name:18s}", " | ".join(res), flush=True)
