"""Dev tool: the short-K pointwise layers on the implicit GEMM (table tiles) vs the streaming kernel (tile_hint 5)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
import ctypes as C
lib, st = _lib.load(), _lib.current_stream()
LAYERS = [("res group0 conv3 64->256 +res", 16, 187, 333, 64, 256, True), ("res group0 shortcut 64->256", 16, 187, 333, 64, 256, False),
          ("res group1 conv3 128->512 +res", 16, 94, 167, 128, 512, True), ("xc entry b1 pw 128->128 +res", 160, 97, 97, 128, 128, True),
          ("xc entry b1 pw 64->128", 160, 193, 193, 64, 128, False), ("xc entry b2 pw 128->256", 160, 97, 97, 128, 256, False)]
for name, n, h, w, cin, cout, res in LAYERS:
    x = ops.NHWC(torch.randn((n, h, w, cin), device="cuda"), c=cin)
    out = ops.NHWC.alloc(n, h, w, cout)
    r = ops.NHWC(torch.randn((n, h, w, cout), device="cuda"), c=cout) if res else None
    pk = ops.pack_conv(torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5, torch.zeros(cout))
    row = []
    for hint in ((64 << 16) | 64, (128 << 16) | 128, (128 << 16) | 64, 5):
        d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU, res=r, tile_hint=hint, stage_k=16, split_k=-1)
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
        row.append(f"{'stream' if hint == 5 else f'{hint >> 16}x{hint & 0xffff}'}: {best:7.1f} us {ops.algorithmic_bytes(d) / best / 1e3:5.0f} GB/s")
    print(f"{name:32s}", " | ".join(row), flush=True)
