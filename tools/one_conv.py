"""Dev tool: one conv layer on the implicit-GEMM kernel, N launches (for rocprofv3 / tools/pmc_kernel.sh).
   python tools/one_conv.py n h w cin cout k [tile_hint_bm tile_hint_bn stage_k]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from premvos_amd import ops, _lib
import ctypes as C
n, h, w, cin, cout, k = (int(v) for v in sys.argv[1:7])
bm, bn, sk = (int(v) for v in sys.argv[7:10]) if len(sys.argv) >= 10 else (128, 128, 16)
x = ops.NHWC(torch.randn((n, h, w, (cin + 3) // 4 * 4), device="cuda"), c=cin)
out = ops.NHWC.alloc(n, h, w, cout)
pk = ops.pack_conv(torch.randn((cout, cin, k, k)) * (2.0 / (k * k * cin)) ** 0.5, torch.zeros(cout))
d = ops.conv_desc(x, pk, out, pad=(k // 2, k // 2), act=ops.ACT_RELU, tile_hint=(bm << 16) | bn, stage_k=sk, split_k=-1)
lib, st = _lib.load(), _lib.current_stream()
for _ in range(3):
    _lib.check(lib.premvos_conv2d_f32(C.byref(d), st))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    lib.premvos_conv2d_f32(C.byref(d), st)
b.record(); b.synchronize()
us = a.elapsed_time(b) * 100
print(f"{us:.1f} us  {2.0 * n * h * w * k * k * cin * cout / us / 1e6:.1f} TF/s")
