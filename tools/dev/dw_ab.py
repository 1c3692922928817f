"""Dev tool (round 5): the depthwise 3x3 kernel alone and followed by its pointwise conv (what the next kernel pays for where the
depthwise output was left), middle-flow / entry-flow shapes of a 160-crop refinement call.  PREMVOS_LIB_PATH selects the build."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import _lib, ops
from premvos_amd.refinement.model import PackedDW
lib, st = _lib.load(), _lib.current_stream()


def timeit(fn, reps=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1000 / reps)
    return sorted(ts)[len(ts) // 2]


for name, n, h, c, cout in (("middle 728 @25x25", 160, 25, 728, 728), ("entry 256 @97x97", 160, 97, 256, 256), ("entry 128 @193x193", 160, 193, 128, 128),
                            ("exit 1536 @25x25", 160, 25, 1536, 1536)):
    g = torch.Generator().manual_seed(1)
    x = ops.NHWC(torch.randn((n, h, h, c), device="cuda"), c=c)
    t = ops.NHWC.alloc(n, h, h, c)
    o = ops.NHWC.alloc(n, h, h, cout)
    bn = {"gamma": torch.ones(c), "beta": torch.zeros(c), "mean": torch.zeros(c), "var": torch.ones(c)}
    k = PackedDW(torch.randn((c, 1, 3, 3), generator=g), bn, 1e-3, "cuda")
    pk = ops.pack_conv(torch.randn((cout, c, 1, 1), generator=g) * (2.0 / c) ** 0.5, torch.zeros(cout))
    d = ops.conv_desc(t, pk, o, act=ops.ACT_NONE)
    ops.autotune([d])
    ws = ops.assign_workspace([d])
    dw = lambda: lib.premvos_dwconv3x3_f32(x.ptr, x.ps, n, h, h, c, k.wgt.data_ptr(), k.bias.data_ptr(), k.c_pad, t.ptr, t.ps, h, h, 1, 1, 1, 1, 1, 0, st)   # noqa: E731
    pw = lambda: lib.premvos_conv2d_f32(C.byref(d), st)   # noqa: E731
    t_dw, t_pw = timeit(dw), timeit(pw)
    t_both = timeit(lambda: (dw(), pw()))
    by = 8.0 * n * h * h * c
    print(f"{name:22s} dw {t_dw:8.1f} us = {by / t_dw / 1e6:6.2f} TB/s | pw alone {t_pw:8.1f} | dw + pw {t_both:8.1f} (sum {t_dw + t_pw:8.1f})", flush=True)
    del x, t, o
    torch.cuda.empty_cache()
