"""Cost-volume kernel at the PWC-Net pyramid levels (B pairs): time, algorithmic HBM bytes, GB/s (dev tool).
   python tools/time_corr.py [B]      algorithmic bytes = f1 + f2 read once, 81 (+C copied) floats per pixel written once;
   the fused warp+corr form also reads the 2-channel flow."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from premvos_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
LEVELS = [(6, 196, 8, 14), (5, 128, 16, 28), (4, 96, 32, 56), (3, 64, 64, 112), (2, 32, 128, 224)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3          # us


for lvl, c, h, w in LEVELS:
    f1 = ops.NHWC(torch.randn((B, h, w, c), device="cuda"))
    f2 = ops.NHWC(torch.randn((B, h, w, c), device="cuda"))
    flow = ops.NHWC(torch.randn((B, h, w, 4), device="cuda"), c=2)
    out = ops.NHWC.alloc(B, h, w, 448 + 81 + c + 4)
    dst = out.slice(448, 81 + c)
    npix = B * h * w
    byt = npix * 4 * (2 * c + 81 + c)
    t = timed(lambda: ops.corr(f1, f2, dst, 4, 0.1, True))
    dig = int(out.buf.view(torch.int32).to(torch.int64).sum().item())      # (same seed, same bits: variants must agree)
    tw = timed(lambda: ops.warp_corr(f1, f2, flow, 1.25, dst, 4, 0.1, True))
    wb = ops.NHWC.alloc(B, h, w, c)
    t2 = timed(lambda: (ops.warp(f2, flow, 1.25, wb), ops.corr(f1, wb, dst, 4, 0.1, True)))
    print(f"level {lvl}: C={c:3d} {h:3d}x{w:3d} x{B}  corr {t:8.1f} us  {byt / t / 1e6:7.2f} TB/s  digest {dig & 0xffffffffffff:012x} | warp+corr fused {tw:8.1f} us "
          f"({(byt + npix * 8) / tw / 1e6:5.2f} TB/s) vs warp, corr {t2:8.1f} us", flush=True)

# the backward bilinear warp on its own (PWCDCNet.warp, PWCNet.py:140-176; flow_ops.hip::warp_kernel): reads the C channels of up to
# four neighbours per pixel (unique: the feature map once) + the 2-channel flow, writes C channels -- algorithmic bytes = 2 C + 2 floats
print()
for lvl, c, h, w in LEVELS[1:]:
    f2 = ops.NHWC(torch.randn((B, h, w, c), device="cuda"))
    flow = ops.NHWC((torch.randn((B, h, w, 4), device="cuda") * 2.0).contiguous(), c=2)
    wb = ops.NHWC.alloc(B, h, w, c)
    byt = B * h * w * 4 * (2 * c + 2)
    t = timed(lambda: ops.warp(f2, flow, 1.25, wb), reps=50)
    print(f"warp level {lvl}: C={c:3d} {h:3d}x{w:3d} x{B}  {t:7.1f} us  {byt / t / 1e6:6.2f} TB/s algorithmic ({byt / 1e6:.1f} MB)", flush=True)
