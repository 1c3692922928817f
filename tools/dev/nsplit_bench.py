"""Dev tool: is a 728-wide pointwise layer faster as 640 + 88 output columns (two launches)?  Best candidate of the tuner's list each."""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
lib, st = _lib.load(), _lib.current_stream()


def best(n, h, w, cin, cout):
    x = ops.NHWC(torch.randn((n, h, w, (cin + 3) // 4 * 4), device="cuda"), c=cin)
    out = ops.NHWC.alloc(n, h, w, cout)
    pk = ops.pack_conv(torch.randn((cout, cin, 1, 1)) * 0.05, torch.zeros(cout))
    d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU)
    res = (1e30, None)
    for cand in ops._candidates(d):
        d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = cand
        ws = ops.assign_workspace([d])
        if lib.premvos_conv2d_f32(C.byref(d), st) != 0:
            continue
        t = 1e30
        for _ in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                lib.premvos_conv2d_f32(C.byref(d), st)
            b.record(); b.synchronize()
            t = min(t, a.elapsed_time(b) * 200)
        if t < res[0]:
            res = (t, cand)
    return res


for n, h, w, cin in ((160, 25, 25, 728), (160, 49, 49, 728), (160, 49, 49, 256)):
    full = best(n, h, w, cin, 728)
    a, b = best(n, h, w, cin, 640), best(n, h, w, cin, 88)
    fl = 2.0 * n * h * w * cin * 728
    print(f"{n}x{h}x{w} K={cin}: N=728 {full[0]:8.1f} us ({fl / full[0] / 1e6:5.1f} TF/s) {full[1]} | 640: {a[0]:8.1f} {a[1]} + 88: {b[0]:8.1f} {b[1]} = {a[0] + b[0]:8.1f} us "
          f"({fl / (a[0] + b[0]) / 1e6:5.1f} TF/s)", flush=True)
