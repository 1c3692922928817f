"""Dev tool: separable conv halves of the refinement net: fp32 (dw + implicit GEMM) vs bf16x3 on the fly vs bf16x3 split path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
import ctypes as C
lib, st = _lib.load(), _lib.current_stream()

def timeit(fn, reps=10):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b) * 1000 / reps)
    return best

for name, n, h, w, cin, cout in (("mid 728->728 M=100k", 160, 25, 25, 728, 728), ("mid 728->728 M=384k", 160, 49, 49, 728, 728),
                                  ("exit 1536->2048", 160, 25, 25, 1536, 2048), ("entry 128->256 97x97", 160, 97, 97, 128, 256),
                                  ("entry 64->128 193x193", 160, 193, 193, 64, 128)):
    x = ops.NHWC.alloc(n, h, w, cin); x.buf.normal_()
    t = ops.NHWC.alloc(n, h, w, cin)
    out = ops.NHWC.alloc(n, h, w, cout)
    wt = torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5
    cpad = (cin + 3) // 4 * 4
    dwk = torch.randn((9, cpad), device="cuda") * 0.3
    b0 = torch.zeros((cpad,), device="cuda")
    def dw(flags):
        _lib.check(lib.premvos_dwconv3x3_f32(x.ptr, x.ps, n, h, w, cin, dwk.data_ptr(), b0.data_ptr(), cpad, t.ptr, t.ps, h, w, 1, 1, 1, 1, 1, flags, st))
    flops = 2.0 * n * h * w * cin * cout
    res = {}
    pk32 = ops.pack_conv(wt, torch.zeros(cout))
    d32 = ops.conv_desc(t, pk32, out, act=ops.ACT_RELU, tile_hint=(128 << 16) | 128, stage_k=16, split_k=-1)
    pk3 = ops.pack_conv(wt, torch.zeros(cout), precision="bf16x3")
    d3 = ops.conv_desc(t, pk3, out, act=ops.ACT_RELU)
    t_dw = timeit(lambda: dw(0)); t_dws = timeit(lambda: dw(_lib.ACT_SPLIT_BF16))
    dw(0)
    t32 = timeit(lambda: lib.premvos_conv2d_f32(C.byref(d32), st))
    t3 = timeit(lambda: lib.premvos_conv2d_f32(C.byref(d3), st))
    dw(_lib.ACT_SPLIT_BF16)
    ts = timeit(lambda: ops.pwconv_bf16x3_split(t, pk3, out, act=ops.ACT_RELU))
    print(f"{name:24s} dw {t_dw:7.1f} us, dw(split) {t_dws:7.1f} us | pw fp32 {t32:8.1f} us {flops / t32 / 1e6:6.1f} TF/s | bf16x3 on the fly {t3:8.1f} us "
          f"{flops / t3 / 1e6:6.1f} | bf16x3 split {ts:8.1f} us {flops / ts / 1e6:6.1f} TF/s-equivalent", flush=True)
