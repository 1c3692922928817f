"""Winograd F(2x2,3x3) conv kernel (csrc/conv_wino_f32.hip) against torch's fp64 convolution and against the implicit-GEMM
kernel on the same descriptor: 3x3 / stride 1, odd map sizes, channel counts that are not multiples of the tiles, fused bias /
residual / activation, output into a channel window."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
FUSED = (0, 2, 3, 4, 5, 6)            # block ids of the slab-free Winograd kernel (tile_hint 3, stage_k)


def _ops():
    from premvos_amd import ops
    return ops


def _nhwc(t, ops, ps=None, coff=0):
    n, c, h, w = t.shape
    x = ops.NHWC.alloc(n, h, w, (ps or (coff + c)))
    x.buf[..., coff:coff + c] = t.permute(0, 2, 3, 1).cuda()
    return x.slice(coff, c)


@pytest.mark.parametrize("n,cin,cout,h,w,act,res", [(1, 64, 64, 17, 23, "relu", True), (2, 117, 128, 32, 40, "leaky", False),
                                                    (1, 256, 256, 46, 83, "relu", False), (3, 20, 36, 9, 5, "none", True),
                                                    (2, 565, 32, 16, 28, "leaky", False), (1, 1024, 1024, 12, 21, "relu", False)])
def test_winograd_matches_torch_and_the_gemm_kernel(n, cin, cout, h, w, act, res):
    ops = _ops()
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    r = torch.randn((n, cout, h, w), generator=g) if res else None
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    if res:
        ref = ref + r.double()
    ref = {"relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.1), "none": lambda t: t}[act](ref)
    a = {"relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY, "none": ops.ACT_NONE}[act]
    pk = ops.pack_conv(wt, b)
    assert pk.wgt_wino is not None and tuple(pk.wgt_wino.shape) == (16, pk.cout_pad, (pk.cin_pad + 15) // 16 * 16)
    xin = _nhwc(x, ops, ps=(cin + 3) // 4 * 4 + 8, coff=4)
    rin = _nhwc(r, ops) if res else None
    outs = {}
    cases = [("wino", 2, 0), ("gemm", (128 << 16) | (128 if cout > 64 else 64 if cout > 32 else 32), 0)]
    cases += [(f"fused{v}", 3, v) for v in FUSED]               # the slab-free kernel, every block / stage depth
    for tag, hint, stage_k in cases:
        out = ops.NHWC.alloc(n, h, w, cout + 8)
        out.buf.fill_(3.0)
        ops.conv2d(xin, pk, out.slice(4, cout), pad=(1, 1), act=a, res=rin, tile_hint=hint, stage_k=stage_k)
        torch.cuda.synchronize()
        assert torch.all(out.buf[..., :4] == 3.0) and torch.all(out.buf[..., 4 + cout:] == 3.0)      # window respected
        outs[tag] = out.slice(4, cout).torch().cpu().double()
    scale = max(1.0, ref.abs().max().item())
    assert (outs["gemm"] - ref).abs().max().item() < 2e-5 * scale
    assert (outs["wino"] - ref).abs().max().item() < 5e-5 * scale            # Winograd rounding: sums before the products
    assert (outs["wino"] - outs["gemm"]).abs().max().item() < 5e-5 * scale
    for v in FUSED:
        assert (outs[f"fused{v}"] - ref).abs().max().item() < 5e-5 * scale, v
        # same products, same K order; only the order of the sixteen component sums differs from the slab version
        assert (outs[f"fused{v}"] - outs["wino"]).abs().max().item() < 2e-5 * scale, v


@pytest.mark.parametrize("n,cin,cout,h,w,dil", [(2, 128, 128, 32, 40, 2), (1, 96, 64, 37, 29, 4), (1, 128, 96, 40, 56, 8), (1, 64, 128, 33, 50, 16)])
def test_atrous_winograd_matches_torch(n, cin, cout, h, w, dil):
    """Dilated 3x3 / pad = dilation layers (PWC-Net's context net, PWCNet.py:125-131) on the slab-free kernel: Winograd on the
    dil x dil sub-lattices, every block."""
    ops = _ops()
    g = torch.Generator().manual_seed(cin + dil)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=dil, dilation=dil), 0.1)
    pk = ops.pack_conv(wt, b)
    xin = _nhwc(x, ops)
    scale = max(1.0, ref.abs().max().item())
    for v in ((3, 5) if cout <= 64 else FUSED):
        out = ops.NHWC.alloc(n, h, w, cout)
        ops.conv2d(xin, pk, out, pad=(dil, dil), dilation=(dil, dil), act=ops.ACT_LEAKY, tile_hint=3, stage_k=v)
        torch.cuda.synchronize()
        assert (out.torch().cpu().double() - ref).abs().max().item() < 5e-5 * scale, v
    d = ops.conv_desc(xin, pk, ops.NHWC.alloc(n, h, w, cout), pad=(dil, dil), dilation=(dil, dil))
    assert any(c[0] == 3 for c in ops._candidates(d)) and not any(c[0] == 2 for c in ops._candidates(d))


@pytest.mark.parametrize("n,cin,cout,h,w,act,res", [(1, 256, 256, 46, 83, "relu", False), (2, 128, 128, 32, 40, "leaky", True),
                                                    (1, 1024, 1024, 12, 21, "relu", False), (3, 130, 132, 9, 5, "none", True),
                                                    (5, 512, 512, 7, 7, "relu", True), (1, 160, 136, 4, 3, "relu", False), (2, 245, 64, 20, 36, "leaky", False),
                                                    (2, 533, 32, 20, 36, "leaky", False), (1, 520, 32, 9, 14, "leaky", True)])     # PWC conv*_4: 32-column GEMM block (Kp % 32 == 0) / 64
def test_winograd_4x4_matches_torch(n, cin, cout, h, w, act, res):
    """F(4x4,3x3) (csrc/conv_wino4_f32.hip: input transform, 36 batched GEMMs, output transform; tile_hint 4) on K-rich layers:
    maps that are not multiples of the 4x4 tile, channel counts off the GEMM tiles, channel windows, fused epilogue."""
    ops = _ops()
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    r = torch.randn((n, cout, h, w), generator=g) if res else None
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    if res:
        ref = ref + r.double()
    ref = {"relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.1), "none": lambda t: t}[act](ref)
    a = {"relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY, "none": ops.ACT_NONE}[act]
    pk = ops.pack_conv(wt, b)
    assert pk.wgt_wino4 is not None and tuple(pk.wgt_wino4.shape) == (36, pk.cout_pad, (pk.cin_pad + 15) // 16 * 16)
    xin = _nhwc(x, ops, ps=(cin + 3) // 4 * 4 + 8, coff=4)
    rin = _nhwc(r, ops) if res else None
    scale = max(1.0, ref.abs().max().item())
    d = ops.conv_desc(xin, pk, ops.NHWC.alloc(n, h, w, cout), pad=(1, 1))
    assert any(c[0] == 4 for c in ops._candidates(d))
    for stage_k, coff, extra in ((0, 4, 8), (64, 0, 0), (16, 0, 4), (80, 4, 4), (0, 2, 3)):      # (0, 2, 3): an unaligned window -> scalar stores
        out = ops.NHWC.alloc(n, h, w, cout + extra)
        out.buf.fill_(3.0)
        ops.conv2d(xin, pk, out.slice(coff, cout), pad=(1, 1), act=a, res=rin, tile_hint=4, stage_k=stage_k)
        torch.cuda.synchronize()
        assert torch.all(out.buf[..., :coff] == 3.0) and torch.all(out.buf[..., coff + cout:] == 3.0)      # window respected
        got = out.slice(coff, cout).torch().cpu().double()
        # the transforms scale by up to 100 / down to 1/576 before the products: ~1e-5 of the output scale (F(2x2,3x3): ~1e-6)
        assert (got - ref).abs().max().item() < 2e-4 * scale, (stage_k, coff)


def test_winograd_is_refused_where_it_does_not_apply():
    from premvos_amd import _lib
    ops = _ops()
    x = ops.NHWC.alloc(1, 8, 8, 16)
    pk = ops.pack_conv(torch.randn((16, 16, 3, 3)), None)
    out = ops.NHWC.alloc(1, 4, 4, 16)
    with pytest.raises(_lib.PremvosError, match="Winograd"):
        ops.conv2d(x, pk, out, stride=(2, 2), pad=(1, 1), tile_hint=2)          # stride 2
    with pytest.raises(_lib.PremvosError, match="Winograd"):
        ops.conv2d(x, pk, out, stride=(2, 2), pad=(1, 1), tile_hint=3)
    out8 = ops.NHWC.alloc(1, 8, 8, 16)
    assert pk.wgt_wino4 is None                   # 16 channels: F(4x4,3x3) is only packed for K- and N-rich layers
    with pytest.raises(_lib.PremvosError, match="4x4"):
        ops.conv2d(x, pk, out8, pad=(1, 1), tile_hint=4)
    x2 = ops.NHWC.alloc(1, 8, 8, 128)
    pk2 = ops.pack_conv(torch.randn((128, 128, 3, 3)) * 0.03, None)
    with pytest.raises(_lib.PremvosError, match="block id"):
        ops.conv2d(x2, pk2, ops.NHWC.alloc(1, 8, 8, 128), pad=(1, 1), tile_hint=4, stage_k=32)
    out1 = ops.NHWC.alloc(1, 8, 8, 16)
    with pytest.raises(_lib.PremvosError, match="block id"):
        ops.conv2d(x, pk, out1, pad=(1, 1), tile_hint=3, stage_k=11)
    pk1 = ops.pack_conv(torch.randn((16, 16, 1, 1)), None)
    assert pk1.wgt_wino is None


def test_kept_winograd_slab_of_a_densenet_block_is_bit_identical():
    """premvos_conv_wino4_slab_f32: a PWC-Net estimator level in miniature (PWCNet.py:201-205: every layer prepends its output
    to the concat buffer and the next layer reads the longer window).  Self-contained F(4x4) per layer against ONE kept slab in
    which every layer transforms only the channels the previous one added: same bits, layer by layer."""
    from premvos_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    n, h, w, od, growth = 2, 18, 27, 149, (128, 128, 96, 64)           # 149 = 81 + 64 + 4 channels under the grown ones (level 3)
    total = sum(growth) + od
    packs, off = [], sum(growth)
    for gch in growth:
        cin = total - off
        packs.append((off, gch, ops.pack_conv(torch.randn((gch, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5,
                                              torch.randn((gch,), generator=g) * 0.1)))
        off -= gch
    base = torch.randn((n, h, w, od), generator=g).cuda()

    def fresh():
        X = ops.NHWC.alloc(n, h, w, total)
        X.buf.zero_()
        X.buf[..., sum(growth):total] = base
        return X

    def descs(X):
        ds = []
        for off, gch, pk in packs:
            d = ops.conv_desc(X.slice(off, total - off), pk, X.slice(off - gch, gch), pad=(1, 1), act=ops.ACT_LEAKY, tile_hint=4)
            ds.append((d, off))
        ws = ops.assign_workspace([d for d, _ in ds])
        return ds, ws

    Xa = fresh()
    da, wsa = descs(Xa)
    for d, _ in da:
        ops.run_desc(d)
    Xb = fresh()
    db, wsb = descs(Xb)
    need, plan = ops.wino4_slab_plan(db)
    assert [(c0, t) for _, _, c0, t in plan] == [(416, 160), (288, 128), (160, 128), (64, 96)]       # first: its whole Kp = 160
    assert plan[0][1] == 576 and need == 36 * n * 5 * 7 * 576                      # pitch: the buffer's 565 channels rounded to 16
    slab = torch.full((need,), float("nan"), device="cuda")                         # nothing stale may be read
    for d, pitch, c0, t_cn in plan:
        ops.run_wino4_slab(d, slab, pitch, c0, t_cn)
    torch.cuda.synchronize()
    assert torch.equal(Xa.buf, Xb.buf) and bool(torch.isfinite(Xb.buf).all())
    # a second pass over the same slab (a replayed launch list) and t_cn = 0 (everything already there): same bits again
    Xb.buf[..., :sum(growth)].zero_()
    for d, pitch, c0, t_cn in plan:
        ops.run_wino4_slab(d, slab, pitch, c0, t_cn)
    d3, pitch, c0, _ = plan[3]
    ops.run_wino4_slab(d3, slab, pitch, c0, 0)
    torch.cuda.synchronize()
    assert torch.equal(Xa.buf, Xb.buf)
    # what the entry point refuses
    d0 = plan[0][0]
    with pytest.raises(_lib.PremvosError, match="do not fit"):
        ops.run_wino4_slab(d0, slab, 560, 416, 128)                                 # window end beyond the pitch
    with pytest.raises(_lib.PremvosError, match="do not fit"):
        ops.run_wino4_slab(d0, slab, 576, 414, 128)                                 # not a multiple of 4
    with pytest.raises(_lib.PremvosError, match="needs"):
        ops.run_wino4_slab(d0, slab[:1000], 576, 416, 128)
    pk1 = ops.pack_conv(torch.randn((16, 16, 3, 3)), None)
    x1 = ops.NHWC.alloc(1, 8, 8, 16)
    d1 = ops.conv_desc(x1, pk1, ops.NHWC.alloc(1, 8, 8, 16), pad=(1, 1))
    with pytest.raises(_lib.PremvosError, match="not a 3x3"):
        ops.run_wino4_slab(d1, slab, 16, 0, 16)
    # a block whose windows do not start at multiples of 16 gets no plan (the layers then run self-contained)
    assert ops.wino4_slab_plan([(db[0][0], 321), (db[1][0], 192)]) is None


@pytest.mark.parametrize("n,cin,cout,h,w,dil", [(2, 128, 128, 32, 40, 2), (1, 128, 96, 37, 29, 4), (1, 128, 128, 40, 56, 8), (1, 160, 64, 33, 50, 16)])
def test_winograd_4x4_on_atrous_layers_matches_torch(n, cin, cout, h, w, dil):
    """PWC-Net's context network (PWCNet.py:266-267: 3x3, dilation 2 / 4 / 8 / 16, pad = dilation) on F(4x4,3x3): the d x d
    interleaved sub-lattices are independent dense problems -- tiles per phase, inputs d pixels apart; maps that are no multiple of
    4 d, every GEMM block, output into a channel window with a residual."""
    ops = _ops()
    g = torch.Generator().manual_seed(cin + dil)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    r = torch.randn((n, cout, h, w), generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=dil, dilation=dil) + r.double(), 0.1)
    pk = ops.pack_conv(wt, b)
    xin, rin = _nhwc(x, ops, ps=cin + 4, coff=4), _nhwc(r, ops)
    d = ops.conv_desc(xin, pk, ops.NHWC.alloc(n, h, w, cout), pad=(dil, dil), dilation=(dil, dil))
    assert any(c[0] == 4 for c in ops._candidates(d))
    scale = max(1.0, ref.abs().max().item())
    for stage_k in (0, 16, 64, 80):
        out = ops.NHWC.alloc(n, h, w, cout + 8)
        out.buf.fill_(3.0)
        ops.conv2d(xin, pk, out.slice(4, cout), pad=(dil, dil), dilation=(dil, dil), act=ops.ACT_LEAKY, res=rin, tile_hint=4, stage_k=stage_k)
        torch.cuda.synchronize()
        assert torch.all(out.buf[..., :4] == 3.0) and torch.all(out.buf[..., 4 + cout:] == 3.0)
        got = out.slice(4, cout).torch().cpu().double()
        assert (got - ref).abs().max().item() < 2e-4 * scale, (stage_k, dil)
