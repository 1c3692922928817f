python - <<'PY'
import sys, os, torch, ctypes as C
sys.path.insert(0, '.')
from premvos_amd import ops, _lib
for name, n, h, w, cin, cout, k in (("predict_flow2", 16, 128, 224, 565, 2, 3), ("logits", 160, 97, 97, 256, 2, 1), ("predict_flow3", 16, 64, 112, 597, 2, 3)):
    x = ops.NHWC(torch.randn((n, h, w, (cin + 3) // 4 * 4), device="cuda"), c=cin)
    out = ops.NHWC.alloc(n, h, w, cout)
    pk = ops.pack_conv(torch.randn((cout, cin, k, k)) * 0.05, torch.zeros(cout))
    d = ops.conv_desc(x, pk, out, pad=(k // 2, k // 2), tile_hint=1)
    lib, st = _lib.load(), _lib.current_stream()
    for _ in range(3): _lib.check(lib.premvos_conv2d_f32(C.byref(d), st))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): lib.premvos_conv2d_f32(C.byref(d), st)
    b.record(); b.synchronize()
    us = a.elapsed_time(b) * 100
    print(f"{name:16s} {us:8.1f} us  input {x.buf.numel() * 4 / us / 1e6:5.2f} TB/s-equiv")
PY
