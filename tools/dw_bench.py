"""Dev tool: depthwise 3x3 kernel on the Xception shapes of a refinement call (160 boxes): time, algorithmic GB/s (in + out)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from premvos_amd import _lib

SHAPES = [("middle 728 @25x25", 160, 25, 25, 728, 1, 1), ("entry3 728 @49->25 s2", 160, 49, 49, 728, 2, 1),
          ("entry2 256 @97x97", 160, 97, 97, 256, 1, 1), ("entry1 128 @193x193", 160, 193, 193, 128, 1, 1),
          ("exit 1536 @25x25 rate2", 160, 25, 25, 1536, 1, 2), ("aspp 2048 @25x25 rate12", 160, 25, 25, 2048, 1, 12),
          ("decoder 304 @97x97", 160, 97, 97, 304, 1, 1)]
lib, st = _lib.load(), _lib.current_stream()
for name, n, h, w, c, s, d in SHAPES:
    ho, wo = -(-h // s), -(-w // s)
    x = torch.randn((n, h, w, c), device="cuda")
    y = torch.empty((n, ho, wo, c), device="cuda")
    wt, b = torch.randn((9, c), device="cuda"), torch.randn((c,), device="cuda")
    pt = max((ho - 1) * s + (3 - 1) * d + 1 - h, 0) // 2

    def run():
        _lib.check(lib.premvos_dwconv3x3_f32(x.data_ptr(), c, n, h, w, c, wt.data_ptr(), b.data_ptr(), c, y.data_ptr(), c, ho, wo, s, d,
                                             pt, pt, 1, 0, st))
    for _ in range(3):
        run()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        run()
    e.record(); e.synchronize()
    us = a.elapsed_time(e) * 100
    gb = (x.numel() + y.numel()) * 4 / 1e9
    print(f"{name:28s} {us:8.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s", flush=True)
