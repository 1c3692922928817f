"""Dev tool: per-tile fixed cost of the fp32 implicit-GEMM kernel -- a 1x1 layer with NO padding waste and whole rounds of
workgroups (M = 98304 = 768 row tiles, N = 768 = 6 column tiles: 4608 workgroups = 6 rounds of 3 per CU), K swept:
time(K) = K / rate_inf + fixed.  The fit says what a persistent / overlapped-epilogue kernel could recover at K = 728."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
import ctypes as C
lib, st = _lib.load(), _lib.current_stream()
n, h, w, cout = 96, 32, 32, 768
pts = []
for cin in (128, 256, 512, 768, 1024, 1536, 3072):
    x = ops.NHWC(torch.randn((n, h, w, cin), device="cuda"), c=cin)
    out = ops.NHWC.alloc(n, h, w, cout)
    pk = ops.pack_conv(torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5, torch.zeros(cout))
    row = []
    for sk in (16, 32):
        d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU, tile_hint=(128 << 16) | 128, stage_k=sk, split_k=-1)
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
        row.append(best)
    pts.append((cin, row))
    print(f"K={cin:5d}  " + "  ".join(f"stage {sk}: {t:8.1f} us {2.0 * n * h * w * cin * cout / t / 1e6:6.1f} TF/s" for sk, t in zip((16, 32), row)), flush=True)
for j, sk in enumerate((16, 32)):
    (k0, r0), (k1, r1) = pts[2], pts[-1]
    slope = (r1[j] - r0[j]) / (k1 - k0)
    fixed = r0[j] - slope * k0
    print(f"stage {sk}: asymptotic rate {2.0 * n * h * w * cout / slope / 1e6:6.1f} TF/s, fixed cost {fixed:6.1f} us per launch = {fixed / slope:5.0f} k-steps "
          f"({100 * fixed / (fixed + 736 * slope):4.1f} % of a K = 736 layer)")
