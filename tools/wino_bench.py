"""Dev tool: Winograd F(2x2,3x3) kernel vs the implicit-GEMM kernel (its best tile) on the 3x3 layers of the pipeline (B = 16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from premvos_amd import ops, _lib

LAYERS = [("rpn 3x3 1024->1024 @46x83", 16, 1024, 1024, 46, 83), ("resnet conv4 256->256 @46x83", 16, 256, 256, 46, 83),
          ("resnet conv3 128->128 @94x167", 16, 128, 128, 94, 167), ("resnet conv2 64->64 @187x333", 16, 64, 64, 187, 333),
          ("pwc dc_conv1 565->128 @128x224", 16, 565, 128, 128, 224), ("pwc conv2_1 245->128 @128x224", 16, 245, 128, 128, 224),
          ("pwc conv2_2 373->96 @128x224", 16, 373, 96, 128, 224), ("pwc conv2_3 469->64 @128x224", 16, 469, 64, 128, 224),
          ("pwc conv2_4 533->32 @128x224", 16, 533, 32, 128, 224), ("pwc conv3_1 277->128 @64x112", 16, 277, 128, 64, 112),
          ("conv5 512->512 @7x7 x1600", 1600, 512, 512, 7, 7)]


def timed(d, reps=5):
    lib, st = _lib.load(), _lib.current_stream()
    import ctypes as C
    for _ in range(2):
        _lib.check(lib.premvos_conv2d_f32(C.byref(d), st))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        lib.premvos_conv2d_f32(C.byref(d), st)
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, n, cin, cout, h, w in LAYERS:
    x = ops.NHWC(torch.randn((n, h, w, (cin + 3) // 4 * 4), device="cuda"), c=cin)
    out = ops.NHWC.alloc(n, h, w, cout)
    pk = ops.pack_conv(torch.randn((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5, torch.zeros(cout))
    flops = 2.0 * n * h * w * 9 * cin * cout
    d = ops.conv_desc(x, pk, out, pad=(1, 1), act=ops.ACT_RELU)
    best, tw, tf, t4 = (1e30, None), (1e30, None), {}, {}
    for cand in ops._candidates(d):
        d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = cand
        ws = ops.assign_workspace([d])
        t = timed(d)
        if cand[0] == 2:
            tw = min(tw, (t, cand[1]))
        elif cand[0] == 3:
            tf[cand[1]] = t
        elif cand[0] == 4:
            t4[cand[1]] = t
        elif t < best[0]:
            best = (t, cand)
    fused = " ".join(f"[{v}] {t:8.1f}" for v, t in sorted(tf.items()))
    tfb = min(tf.values()) if tf else float("nan")
    print(f"{name:34s} gemm {best[0]:9.1f} us {flops / best[0] / 1e6:6.1f} TF/s | slabs {tw[0]:9.1f} us (bm {tw[1] or 128}) {flops / tw[0] / 1e6:6.1f} | "
          f"fused {fused} us -> {flops / tfb / 1e6:6.1f} TF/s-equiv  x{tw[0] / tfb:.2f} vs slabs"
          + (" | F(4x4) " + " ".join(f"[{v or 128}] {t:8.1f}" for v, t in sorted(t4.items())) + f" us -> {flops / min(t4.values()) / 1e6:6.1f} TF/s-equiv"
             f"  x{min(best[0], tw[0], tfb) / min(t4.values()):.2f} vs the best other" if t4 else ""), flush=True)
