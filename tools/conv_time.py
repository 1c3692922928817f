"""Dev tool: TF/s of a few conv shapes under forced (tile, stage) configurations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from premvos_amd import ops
SHAPES = {"pwc_conv2_4": (16, 128, 224, 533, 32, 3), "pwc_conv2_3": (16, 128, 224, 469, 64, 3), "res_g0_c1": (16, 187, 333, 256, 64, 1),
          "res_g0_c2": (16, 187, 333, 64, 64, 3),
"pw1024": (8, 128, 128, 1024, 1024, 1), "mid728x4": (80, 25, 25, 728, 728, 1), "c3x3_256": (4, 187, 333, 256, 256, 3),
          "res_1024_256": (4, 47, 84, 1024, 256, 1)}
for name, (n, h, w, cin, cout, k) in SHAPES.items():
    x = ops.NHWC.alloc(n, h, w, cin); x.buf.normal_()
    pk = ops.pack_conv(torch.randn(cout, cin, k, k) * 0.05, torch.zeros(cout), precision=os.environ.get("PREC", "fp32"))
    out = ops.NHWC.alloc(n, h, w, cout)
    fl = 2.0 * n * h * w * cin * cout * k * k
    res = []
    for tile, st in (((128, 128), 16), ((128, 64), 16), ((256, 64), 16), ((128, 32), 16), ((256, 32), 16), ((64, 64), 16)):
        if tile[1] > max(32, -(-cout // 32) * 32) and tile[1] != 128:
            continue
        d = ops.conv_desc(x, pk, out, pad=(k // 2, k // 2), act=ops.ACT_RELU, tile_hint=(tile[0] << 16) | tile[1], stage_k=st, split_k=-1)
        for _ in range(3): ops.run_desc(d)
        torch.cuda.synchronize()
        reps = max(5, int(0.25e12 * 100 / fl / 100))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): ops.run_desc(d)
        b.record(); b.synchronize()
        res.append(f"{tile[0]}x{tile[1]}/{st}: {fl * reps / (a.elapsed_time(b) * 1e-3) / 1e12:6.1f}")
    print(f"  {name:14s} " + "   ".join(res))
