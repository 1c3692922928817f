"""csrc/conv_bf16x3_s8.hip (round 4): convolution as an implicit GEMM on the bf16 matrix pipe, hi.hi + hi.lo + lo.hi with fp32
accumulation, on activations resident in the split layout S8 and staged by LDS-DMA.  Checked against the fp64 convolution of the
same fp32 tensors (torch CPU) at the accuracy class the mode promises (~2^-16 relative per product, fp32 accumulate): every tile
configuration, 1x1 / 3x3 / strided / dilated / asymmetric padding, channel counts with a partial last 32-channel block, ragged M,
residual + activation epilogues, fp32 and S8 outputs (the S8 output decodes to the fp32 output to bf16x2 precision)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from premvos_amd import ops
    return ops


def _nhwc(x, ops):
    n, c, h, w = x.shape
    v = ops.NHWC.alloc(n, h, w, c, ps=(c + 7) // 8 * 8)
    v.buf[..., :c] = x.permute(0, 2, 3, 1).cuda()
    return v


def _to_s8(x, ops):
    src = _nhwc(x, ops)
    dst = ops.NHWC.alloc_s8(*x.shape[:1], x.shape[2], x.shape[3], x.shape[1])
    ops.split8(src, dst)
    return dst


CASES = [
    # cin, cout, k, stride, dil, pad(t,l), n, h, w, res, act
    (728, 728, 1, 1, 1, (0, 0), 2, 25, 25, False, "relu"),          # Xception middle flow: K = 22.75 blocks, N = 2.84 tiles
    (64, 256, 1, 1, 1, (0, 0), 1, 37, 41, True, "relu"),            # ResNet conv3 + residual, ragged M
    (24, 40, 1, 1, 1, (0, 0), 1, 9, 7, False, "none"),              # partial channel block, cout < tile
    (256, 256, 3, 1, 1, (1, 1), 2, 23, 31, False, "relu"),          # ResNet conv2 3x3
    (64, 64, 3, 2, 1, (0, 0), 1, 30, 33, False, "relu"),            # pad [0,1] + VALID stride 2 (basemodel.py:54-56): implicit bottom/right pad
    (32, 96, 3, 1, 4, (4, 4), 1, 20, 24, False, "leaky"),           # atrous (PWC-Net context net)
    (1024, 512, 1, 2, 1, (0, 0), 3, 14, 14, False, "none"),         # strided 1x1 shortcut
    (120, 128, 3, 1, 1, (1, 1), 1, 16, 28, False, "leaky"),         # PWC estimator-like: cin % 32 != 0
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("tile", list(range(12)))
def test_conv_s8_matches_the_fp64_convolution(case, tile):
    ops = _ops()
    cin, cout, k, stride, dil, pad, n, h, w, res, act = case
    g = torch.Generator().manual_seed(cin * 7 + cout + k + tile)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn((cout,), generator=g)
    keff = dil * (k - 1) + 1
    if stride == 2 and k == 3:
        ho, wo = (h + 1 - keff) // 2 + 1, (w + 1 - keff) // 2 + 1       # pad [0, 1]
        xp = F.pad(x.double(), (0, 1, 0, 1))
        ref = F.conv2d(xp, wt.double(), b.double(), stride=2)
    else:
        ho, wo = (h + 2 * pad[0] - keff) // stride + 1, (w + 2 * pad[1] - keff) // stride + 1
        ref = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=pad, dilation=dil)
    assert ref.shape[2:] == (ho, wo)
    r = torch.randn((n, cout, ho, wo), generator=g) if res else None
    if res:
        ref = ref + r.double()
    ref = {"none": lambda v: v, "relu": F.relu, "leaky": lambda v: F.leaky_relu(v, 0.1)}[act](ref)
    xin = _to_s8(x, ops)
    # the S8 image holds what the kernel multiplies: hi + lo == x to ~2^-17 relative
    assert (xin.torch().cpu() - x).abs().max().item() <= 2.0 ** -15 * x.abs().max().item()
    pk = ops.pack_conv_s8(wt, b)
    out = ops.NHWC.alloc(n, ho, wo, cout, ps=(cout + 7) // 8 * 8)
    out_s8 = ops.NHWC.alloc_s8(n, ho, wo, cout)
    a = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY}[act]
    ops.conv_s8(xin, pk, out, out_s8, tile=tile, stride=(stride, stride), dilation=(dil, dil), pad=pad, act=a,
                res=_nhwc(r, ops) if res else None)
    torch.cuda.synchronize()
    got = out.torch().cpu().double()
    scale = max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() < 3e-5 * scale, (case, tile, (got - ref).abs().max().item())
    got8 = out_s8.torch().cpu().double()
    assert (got8 - got).abs().max().item() <= 2.0 ** -15 * scale
    # S8 output only / fp32 output only give the same numbers
    o2 = ops.NHWC.alloc_s8(n, ho, wo, cout)
    ops.conv_s8(xin, pk, None, o2, tile=tile, stride=(stride, stride), dilation=(dil, dil), pad=pad, act=a, res=_nhwc(r, ops) if res else None)
    torch.cuda.synchronize()
    assert torch.equal(o2.buf, out_s8.buf)


def test_conv_s8_tiles_are_bit_identical_and_chain():
    """Every tile adds an output's products in the same order (32-channel blocks in ascending k; lo.hi, hi.lo, hi.hi per 16-deep
    step): the tile is an order-neutral knob.  And a 1x1 -> 3x3 -> 1x1 chain through S8 buffers only (no fp32 activation in
    between) stays at the mode's accuracy."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 256, 19, 27), generator=g)
    w1 = torch.randn((64, 256, 1, 1), generator=g) * (2.0 / 256) ** 0.5
    w2 = torch.randn((64, 64, 3, 3), generator=g) * (2.0 / 576) ** 0.5
    w3 = torch.randn((256, 64, 1, 1), generator=g) * (1.0 / 64) ** 0.5
    xin = _to_s8(x, ops)
    outs = []
    for tile in range(12):
        o = ops.NHWC.alloc(2, 19, 27, 64)
        ops.conv_s8(xin, ops.pack_conv_s8(w1, None), o, None, tile=tile, act=ops.ACT_RELU)
        torch.cuda.synchronize()
        outs.append(o.buf.clone())
    assert all(torch.equal(outs[0], t) for t in outs[1:])
    t1, t2 = ops.NHWC.alloc_s8(2, 19, 27, 64), ops.NHWC.alloc_s8(2, 19, 27, 64)
    y = ops.NHWC.alloc(2, 19, 27, 256)
    ops.conv_s8(xin, ops.pack_conv_s8(w1, None), None, t1, act=ops.ACT_RELU)
    ops.conv_s8(t1, ops.pack_conv_s8(w2, None), None, t2, pad=(1, 1), act=ops.ACT_RELU)
    ops.conv_s8(t2, ops.pack_conv_s8(w3, None), y, None, res=_nhwc(x, ops), act=ops.ACT_RELU)
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(F.relu(F.conv2d(F.relu(F.conv2d(x.double(), w1.double())), w2.double(), padding=1)), w3.double()) + x.double())
    assert (y.torch().cpu().double() - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item())
    # the same block with the residual read from the S8 tensor (hi + lo) and an S8-only output: what the ResNet chains run
    y8 = ops.NHWC.alloc_s8(2, 19, 27, 256)
    ops.conv_s8(t2, ops.pack_conv_s8(w3, None), None, y8, res_s8=xin, act=ops.ACT_RELU)
    torch.cuda.synchronize()
    assert (y8.torch().cpu().double() - ref).abs().max().item() < 8e-5 * max(1.0, ref.abs().max().item())


def test_conv_s8_refuses_what_it_does_not_cover():
    ops = _ops()
    from premvos_amd import _lib
    x = ops.NHWC.alloc_s8(1, 8, 8, 32)
    pk = ops.pack_conv_s8(torch.randn(16, 32, 1, 1), None)
    o = ops.NHWC.alloc(1, 8, 8, 16)
    with pytest.raises(_lib.PremvosError, match="unknown tile"):
        ops.conv_s8(x, pk, o, None, tile=99)
    with pytest.raises(AssertionError):
        ops.conv_s8(ops.NHWC.alloc(1, 8, 8, 32), pk, o, None)            # an fp32 buffer is not an S8 operand
    with pytest.raises(AssertionError):
        ops.conv2d(x, ops.pack_conv(torch.randn(16, 32, 1, 1), None), o)  # ... and the fp32 kernels refuse an S8 buffer
