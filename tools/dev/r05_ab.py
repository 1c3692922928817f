"""Dev tool (round 5): 128x128 / 16-deep implicit GEMM on the pipeline's big pointwise shapes, time + md5 of the output
(run once per library build: PREMVOS_LIB_PATH=...; compare the md5 columns across builds for bit-identity)."""
import os, sys, hashlib, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
lib, st = _lib.load(), _lib.current_stream()
HINT = int(os.environ.get("AB_HINT", str((128 << 16) | 128)))
LAYERS = [("mid 728->728 25", 160, 25, 25, 728, 728, False), ("mid 728->728 49", 160, 49, 49, 728, 728, False), ("exit 728->1024", 160, 25, 25, 728, 1024, False),
          ("exit 1536->2048", 160, 25, 25, 1536, 2048, False), ("entry 256->728 49", 160, 49, 49, 256, 728, False),
          ("res g2 conv3 256->1024+res", 16, 47, 84, 256, 1024, True), ("res g2 conv1 1024->256", 16, 47, 84, 1024, 256, False),
          ("res g3 conv1 2048->512 roi", 1600, 7, 7, 2048, 512, False), ("ideal 1024->1024", 8, 128, 128, 1024, 1024, False), ("sweep K=256", 96, 32, 32, 256, 768, False)]
tot = 0.0
for name, n, h, w, cin, cout, res in LAYERS:
    torch.manual_seed(1)
    x = ops.NHWC(torch.randn((n, h, w, (cin + 3) // 4 * 4), device="cuda"), c=cin)
    pk = ops.pack_conv(torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5, torch.randn(cout) * 0.1)
    r = ops.NHWC(torch.randn((n, h, w, cout), device="cuda"), c=cout) if res else None
    out = ops.NHWC.alloc(n, h, w, cout)
    d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU, res=r, tile_hint=HINT, stage_k=16, split_k=-1)
    for _ in range(5):
        _lib.check(lib.premvos_conv2d_f32(C.byref(d), st))
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            lib.premvos_conv2d_f32(C.byref(d), st)
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) * 100)
    tot += best
    md5 = hashlib.md5(out.buf.cpu().numpy().tobytes()).hexdigest()[:10]
    print(f"{name:30s} {best:8.1f} us {2.0 * n * h * w * cin * cout / best / 1e6:6.1f} TF/s  md5 {md5}", flush=True)
print(f"total {tot:.1f} us")
