"""Dev tool (round 5): the 128x256 tile (four 64x128 waves, two workgroups per CU; -DPV_DEV_TILES build) against the 128x128 tile on
shapes where BOTH have whole rounds of workgroups (768 / 512 slots): is its LOOP better once tile rounds are out of the picture?"""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
lib, st = _lib.load(), _lib.current_stream()
TILES = [("128x128", (128 << 16) | 128), ("128x256", (128 << 16) | 256), ("256x128w4", (256 << 16) | 129)]
# M = 1536 row tiles of 128: 128x128 with N = 1024: 1536 x 8 = 12288 = 16.0 rounds of 768; 128x256: 1536 x 4 = 6144 = 12.0 rounds of 512;
# 256x128w4: 768 x 8 = 6144 = 12.0 rounds of 512
SHAPES = [("K=256", 192, 32, 32, 256, 1024), ("K=768", 192, 32, 32, 768, 1024), ("K=1536", 192, 32, 32, 1536, 1024), ("K=3072", 192, 32, 32, 3072, 1024)]
for name, n, h, w, cin, cout in SHAPES:
    x = ops.NHWC(torch.randn((n, h, w, cin), device="cuda"), c=cin)
    pk = ops.pack_conv(torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5, torch.randn(cout) * 0.1)
    ref, res = None, []
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
        ref = o if ref is None else ref
        res.append(f"{tname}: {best:7.1f} us {2.0 * n * h * w * cin * cout / best / 1e6:6.1f}{'' if torch.equal(o, ref) else ' MISMATCH'}")
    print(f"{name:8s} M=196608 N=1024 ", " | ".join(res), flush=True)
