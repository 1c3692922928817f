import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from premvos_amd import ops
for name, n, h, w, cin, cout, k in [("xc_mid 728->728", 20, 25, 25, 728, 728, 1), ("g2_c1 1024->256 B4", 4, 46, 83, 1024, 256, 1), ("g2_c2 3x3 256 B4", 4, 46, 83, 256, 256, 3), ("g2_c3 256->1024 B4", 4, 46, 83, 256, 1024, 1), ("c5 3x3 512 x400", 400, 7, 7, 512, 512, 3)]:
    x = ops.NHWC.alloc(n, h, w, cin); x.buf.normal_()
    pk = ops.pack_conv(torch.randn(cout, cin, k, k) * 0.05, torch.zeros(cout))
    out = ops.NHWC.alloc(n, h, w, cout)
    res = []
    for bm, bn in ((128, 128), (64, 128), (128, 64), (64, 64)):
        for sk in (-1, 2, 3):
            for st in (16, 32):
                if st == 32 and (bm, bn) == (64, 64): continue
                try:
                    kw = dict(pad=(k // 2,) * 2, act=ops.ACT_RELU, tile_hint=(bm << 16) | bn, split_k=sk, stage_k=st)
                    for _ in range(3): ops.conv2d(x, pk, out, **kw)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(6): ops.conv2d(x, pk, out, **kw)
                    b.record(); torch.cuda.synchronize()
                    ms = a.elapsed_time(b) / 6
                    res.append((2.0 * n * h * w * k * k * cin * cout / ms / 1e9, f"{bm}x{bn}/sk{sk}/kb{st}"))
                except Exception as e:
                    pass
    res.sort(reverse=True)
    print(name, " | ".join(f"{t}: {v:.1f}" for v, t in res[:6]), "| worst", f"{res[-1][1]}: {res[-1][0]:.1f}", flush=True)
