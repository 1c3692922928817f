"""Dev tool: the ResNet-101 conv4 1x1 layers (46 per net and step) on every tile / stage depth; same-box A/B via PREMVOS_LIB_PATH."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
import ctypes as C
lib, st = _lib.load(), _lib.current_stream()
for name, n, h, w, cin, cout, res in [("conv1 1024->256", 16, 46, 83, 1024, 256, False), ("conv3 256->1024 +res", 16, 46, 83, 256, 1024, True),
                                       ("roi conv1 2048->512", 1600, 7, 7, 2048, 512, False), ("roi conv3 512->2048 +res", 1600, 7, 7, 512, 2048, True)]:
    x = ops.NHWC(torch.randn((n, h, w, cin), device="cuda"), c=cin)
    out = ops.NHWC.alloc(n, h, w, cout)
    r = ops.NHWC(torch.randn((n, h, w, cout), device="cuda"), c=cout) if res else None
    pk = ops.pack_conv(torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5, torch.zeros(cout))
    d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU, res=r)
    cands = [c for c in ops._candidates(d) if c[0] > 4]
    need = 0
    for c in cands:
        d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = c
        need = max(need, ops.workspace_bytes(d))
    ws = torch.empty(need // 4 + 1, dtype=torch.float32, device="cuda")
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    rows = []
    for c in cands:
        d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = c
        if lib.premvos_conv2d_f32(C.byref(d), st) != 0:
            continue
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                lib.premvos_conv2d_f32(C.byref(d), st)
            b.record(); b.synchronize()
            best = min(best, a.elapsed_time(b) * 100)
        rows.append((best, c))
    rows.sort()
    fl = 2.0 * n * h * w * cin * cout
    print(name, " | ".join(f"{c[0] >> 16}x{c[0] & 0xffff}/{c[1]} sk{c[2]} tail{c[3]}x{c[4]}: {t:6.1f} us {fl / t / 1e6:5.1f}" for t, c in rows[:6]), flush=True)
