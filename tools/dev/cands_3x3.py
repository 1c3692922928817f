"""Dev tool: every candidate configuration of one 3x3 / stride-1 layer, timed: python tools/dev/cands_3x3.py n h w cin cout"""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
n, h, w, cin, cout = (int(v) for v in sys.argv[1:6])
x = ops.NHWC(torch.randn((n, h, w, (cin + 3) // 4 * 4), device="cuda"), c=cin)
out = ops.NHWC.alloc(n, h, w, cout)
pk = ops.pack_conv(torch.randn((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5, torch.zeros(cout))
d = ops.conv_desc(x, pk, out, pad=(1, 1), act=ops.ACT_LEAKY)
lib, st = _lib.load(), _lib.current_stream()
rows = []
for cand in ops._candidates(d):
    d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = cand
    ws = ops.assign_workspace([d])
    try:
        for _ in range(3):
            _lib.check(lib.premvos_conv2d_f32(C.byref(d), st), "conv")
    except Exception as e:
        continue
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        lib.premvos_conv2d_f32(C.byref(d), st)
    b.record(); b.synchronize()
    rows.append((a.elapsed_time(b) * 100, cand))
rows.sort()
for us, cand in rows[:8]:
    print(f"{us:9.1f} us  {2.0 * n * h * w * 9 * cin * cout / us / 1e6:7.1f} TF/s-eq  {cand}")
