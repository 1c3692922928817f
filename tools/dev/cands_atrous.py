"""Dev tool: Winograd candidates of one atrous 3x3 layer, timed: python tools/dev/cands_atrous.py n h w cin cout dilation"""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
n, h, w, cin, cout, dl = (int(v) for v in sys.argv[1:7])
x = ops.NHWC(torch.randn((n, h, w, cin), device="cuda"), c=cin)
out = ops.NHWC.alloc(n, h, w, cout)
pk = ops.pack_conv(torch.randn((cout, cin, 3, 3)) * 0.03, torch.zeros(cout))
d = ops.conv_desc(x, pk, out, pad=(dl, dl), dilation=(dl, dl), act=ops.ACT_LEAKY)
lib, st = _lib.load(), _lib.current_stream()
rows = []
for cand in ops._candidates(d):
    if cand[0] not in (3, 4):
        continue
    d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = cand
    ws = ops.assign_workspace([d])
    for _ in range(3):
        _lib.check(lib.premvos_conv2d_f32(C.byref(d), st), "conv")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        lib.premvos_conv2d_f32(C.byref(d), st)
    b.record(); b.synchronize()
    rows.append((a.elapsed_time(b) * 100, cand))
for us, c in sorted(rows)[:7]:
    print(f"{us:8.1f} us {2.0 * n * h * w * 9 * cin * cout / us / 1e6:7.1f} TF/s-eq {c}")
