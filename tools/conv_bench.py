"""Dev tool: micro-benchmark of premvos_conv2d_f32 on representative layer shapes (TF/s per shape and tile)."""
import sys, torch, itertools
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from premvos_amd import ops

SHAPES = [  # name, n, h, w, cin, cout, k, stride, dil
    ("xc_mid_pw 728->728 M12.5k", 20, 25, 25, 728, 728, 1, 1, 1),
    ("xc_exit_pw 1536->2048", 20, 25, 25, 1536, 2048, 1, 1, 1),
    ("xc_entry_pw 128->128 M745k", 20, 193, 193, 128, 128, 1, 1, 1),
    ("xc_b2_pw 256->256 M188k", 20, 97, 97, 256, 256, 1, 1, 1),
    ("rn_g2_c1 1024->256 B4", 4, 46, 83, 1024, 256, 1, 1, 1),
    ("rn_g2_c2 3x3 256->256 B4", 4, 46, 83, 256, 256, 3, 1, 1),
    ("rn_g2_c3 256->1024 B4", 4, 46, 83, 256, 1024, 1, 1, 1),
    ("rn_g2_c2 3x3 256 B1", 1, 46, 83, 256, 256, 3, 1, 1),
    ("rpn 3x3 1024->1024 B4", 4, 46, 83, 1024, 1024, 3, 1, 1),
    ("c5 3x3 512->512 7x7x400", 400, 7, 7, 512, 512, 3, 1, 1),
    ("c5 1x1 2048->512 7x7x400", 400, 7, 7, 2048, 512, 1, 1, 1),
    ("pwc dc1 3x3 565->128 B4", 4, 128, 224, 565, 128, 3, 1, 1),
    ("pwc L2 3x3 341->96 B4", 4, 128, 224, 341, 96, 3, 1, 1),
    ("pwc L5 3x3 661->128 B4 16x28", 4, 16, 28, 597, 128, 3, 1, 1),
    ("big 3x3 256->256 M262k", 4, 256, 256, 256, 256, 3, 1, 1),
]
import os
PRECS = os.environ.get('PRECS', 'fp32').split(',')
hints = [0] + [int(h) for h in sys.argv[1:]]
STAGES = [int(x) for x in os.environ.get('STAGES', '0').split(',')]
for name, n, h, w, cin, cout, k, s, d in SHAPES:
    x = ops.NHWC.alloc(n, h, w, cin); x.buf.normal_()
    wt = torch.randn(cout, cin, k, k) * 0.05
    out = ops.NHWC.alloc(n, h, w, cout)
    res = []
    for prec in PRECS:
      pk = ops.pack_conv(wt, torch.zeros(cout), precision=prec)
      for hint, stage in [(h_, s_) for h_ in hints for s_ in STAGES]:
          try:
              for _ in range(2): ops.conv2d(x, pk, out, pad=(d * (k // 2),) * 2, dilation=(d, d), act=ops.ACT_RELU, tile_hint=hint, stage_k=stage)
          except Exception as e:
              res.append("  n/a"); continue
          a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          reps = 5
          a.record()
          for _ in range(reps): ops.conv2d(x, pk, out, pad=(d * (k // 2),) * 2, dilation=(d, d), act=ops.ACT_RELU, tile_hint=hint, stage_k=stage)
          b.record(); torch.cuda.synchronize()
          ms = a.elapsed_time(b) / reps
          fl = 2.0 * n * h * w * k * k * cin * cout
          res.append(f"{fl/ms/1e9:6.1f}")
    print(f"{name:32s} M={n*h*w:7d} K={k*k*cin:5d} N={cout:5d} | " + " ".join(res), flush=True)
