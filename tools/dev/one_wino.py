"""Dev tool: one 3x3 layer on a chosen Winograd variant, N launches (for tools/pmc_kernel.sh).
   python tools/dev/one_wino.py n h w cin cout tile_hint stage_k"""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
n, h, w, cin, cout, hint, sk = (int(v) for v in sys.argv[1:8])
x = ops.NHWC(torch.randn((n, h, w, (cin + 3) // 4 * 4), device="cuda"), c=cin)
out = ops.NHWC.alloc(n, h, w, cout)
pk = ops.pack_conv(torch.randn((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5, torch.zeros(cout))
d = ops.conv_desc(x, pk, out, pad=(1, 1), act=ops.ACT_RELU, tile_hint=hint, stage_k=sk, split_k=-1)
ws = ops.assign_workspace([d])
lib, st = _lib.load(), _lib.current_stream()
for _ in range(3):
    _lib.check(lib.premvos_conv2d_f32(C.byref(d), st))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    lib.premvos_conv2d_f32(C.byref(d), st)
b.record(); b.synchronize()
us = a.elapsed_time(b) * 100
print(f"{us:.1f} us  {2.0 * n * h * w * 9 * cin * cout / us / 1e6:.1f} TF/s-equivalent")
