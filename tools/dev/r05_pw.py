"""Dev tool (round 5): the persistent pointwise kernel (tile_hint 6) against the 128x128 implicit GEMM: bit-identity + time."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from premvos_amd import ops, _lib
lib, st = _lib.load(), _lib.current_stream()
# name, n, h, w, cin, cout, stride, res, act
SHAPES = [("mid 728->728", 160, 25, 25, 728, 728, 1, False, ops.ACT_NONE), ("mid 728->728 49", 160, 49, 49, 728, 728, 1, False, ops.ACT_NONE),
          ("exit 728->1024", 160, 25, 25, 728, 1024, 1, False, ops.ACT_NONE), ("exit 1536->2048", 160, 25, 25, 1536, 2048, 1, False, ops.ACT_RELU),
          ("entry 256->728 49", 160, 49, 49, 256, 728, 1, False, ops.ACT_NONE), ("res g2 conv3 256->1024+res", 16, 47, 84, 256, 1024, 1, True, ops.ACT_RELU),
          ("res g2 conv1 1024->256", 16, 47, 84, 1024, 256, 1, False, ops.ACT_RELU), ("res g1 conv3 128->512+res", 16, 94, 167, 128, 512, 1, True, ops.ACT_RELU),
          ("res g1 conv1 512->128", 16, 94, 167, 512, 128, 1, False, ops.ACT_RELU), ("shortcut s2 256->512", 16, 188, 334, 256, 512, 2, False, ops.ACT_NONE),
          ("sweep K=3072", 96, 32, 32, 3072, 768, 1, False, ops.ACT_RELU), ("sweep K=256", 96, 32, 32, 256, 768, 1, False, ops.ACT_RELU),
          ("aspp 2048->256", 160, 25, 25, 2048, 256, 1, False, ops.ACT_RELU), ("decoder 304->48ish", 160, 97, 97, 256, 48, 1, False, ops.ACT_RELU),
          ("ragged M", 3, 37, 41, 200, 136, 1, True, ops.ACT_RELU)]
only = os.environ.get("PW_ONLY")
for name, n, h, w, cin, cout, sd, res, act in SHAPES:
    if only and only not in name:
        continue
    ho, wo = (h + sd - 1) // sd, (w + sd - 1) // sd
    x = ops.NHWC(torch.randn((n, h, w, (cin + 3) // 4 * 4), device="cuda"), c=cin)
    pk = ops.pack_conv(torch.randn((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5, torch.randn(cout) * 0.1)
    r = ops.NHWC(torch.randn((n, ho, wo, cout), device="cuda"), c=cout) if res else None
    m = n * ho * wo
    mt, nt = -(-m // 128), -(-cout // 128)
    G = 768
    main_rows = (mt * nt // G) * G // nt
    tail_rows = mt - main_rows
    tail_tiles = tail_rows * nt
    cfgs = [("igemm", (128 << 16) | 128, 0, 0), ("pw", 6, 0, 0)]
    if main_rows > 0 and 0 < tail_tiles <= 0.6 * G:
        ts = min(max(2, round(G / tail_tiles)), 16, pk.k_pad // 64)
        if ts > 1:
            cfgs += [("igemm+tail", (128 << 16) | 128, tail_rows, ts), ("pw+tail", 6, tail_rows, ts)]
    outs, line = {}, []
    for cname, hint, tr, ts in cfgs:
        out = ops.NHWC.alloc(n, ho, wo, cout)
        out.buf.fill_(float("nan"))
        d = ops.conv_desc(x, pk, out, stride=(sd, sd), act=act, res=r, tile_hint=hint, stage_k=16, split_k=-1)
        d.tail_m_tiles, d.tail_split_k = tr, ts
        d.slope = float(os.environ.get("PW_STAGGER", "0"))
        need = ops.workspace_bytes(d)
        ws = torch.empty(max(need // 4, 1), dtype=torch.float32, device="cuda")
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        for _ in range(3):
            _lib.check(lib.premvos_conv2d_f32(C.byref(d), st))
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                lib.premvos_conv2d_f32(C.byref(d), st)
            b.record(); b.synchronize()
            best = min(best, a.elapsed_time(b) * 100)
        outs[cname] = out.buf.clone()
        line.append(f"{cname}: {best:7.1f} us {2.0 * m * cin * cout / best / 1e6:6.1f}")
    ok1 = torch.equal(outs["igemm"], outs["pw"])
    ok2 = ("pw+tail" not in outs) or torch.equal(outs["igemm+tail"], outs["pw+tail"])
    nan = any(bool(torch.isnan(o[..., :cout]).any()) for o in outs.values())
    print(f"{name:28s} tiles {mt * nt:5d} " + " | ".join(line) + f" | bit-identical: {ok1} {ok2}{' NAN!' if nan else ''}", flush=True)
