"""csrc/conv_pwdma_f32.hip (premvos_conv2d_f32, tile_hint 6): pointwise layers with LDS-DMA staged operands give the SAME BITS as the
implicit GEMM's 128 x 128 tile (same products in the same order) -- ragged M, column tiles beyond cout, K that ends inside a stage
(728 = 45.5 x 16: the zero page and the skipped trailing group), stride 2, residual, every activation, channel windows."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # n, h, w, cin, cout, stride, res, act, note
    (2, 25, 25, 728, 728, 1, True, 0, "middle flow"),
    (3, 17, 23, 1024, 256, 1, False, 1, "ragged M"),
    (2, 33, 31, 256, 1024, 1, True, 1, "resnet conv3"),
    (2, 30, 40, 36, 132, 1, False, 2, "K = 36 -> k_pad 48, cout 132"),
    (2, 31, 29, 512, 128, 2, False, 1, "stride 2 shortcut"),
    (1, 9, 11, 2048, 88, 1, False, 3, "narrow single tile, sigmoid"),
    (4, 25, 25, 1536, 2048, 1, False, 1, "exit flow"),
    (1, 5, 7, 40, 72, 1, True, 0, "tiny"),
]


@pytest.mark.parametrize("n,h,w,cin,cout,stride,res,act,note", CASES)
def test_pwdma_is_bit_identical_to_the_implicit_gemm(n, h, w, cin, cout, stride, res, act, note):
    from premvos_amd import _lib, ops
    lib, st = _lib.load(), _lib.current_stream()
    g = torch.Generator().manual_seed(cin * 7 + cout)
    ps = (cin + 3) // 4 * 4 + 8                                   # a channel window of a wider buffer
    xb = torch.randn((n, h, w, ps), generator=g).cuda()
    xb[..., 4 + cin:] = float("nan")                                # whatever lies behind the window's channels must never be read into a sum
    x = ops.NHWC(xb, c=cin, coff=4)
    pk = ops.pack_conv(torch.randn((cout, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5, torch.randn(cout, generator=g) * 0.1)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    r = ops.NHWC(torch.randn((n, ho, wo, cout), generator=g).cuda(), c=cout) if res else None
    outs = []
    for hint in ((128 << 16) | 128, 6):
        out = ops.NHWC(torch.full((n, ho, wo, cout + 4), -7.0, device="cuda"), c=cout)
        d = ops.conv_desc(x, pk, out, stride=(stride, stride), act=act, slope=0.1, res=r, tile_hint=hint, stage_k=16, split_k=-1)
        if hint == 6:
            assert ops.pwdma_applicable(d)
        _lib.check(lib.premvos_conv2d_f32(C.byref(d), st), note)
        torch.cuda.synchronize()
        outs.append(out.buf.clone())
    assert not torch.isnan(outs[0]).any()
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), note       # bit patterns, signed zeros included
    assert bool((outs[1][..., cout:] == -7.0).all())                                     # nothing written beyond the channel window


def test_pwdma_refuses_what_it_cannot_run():
    from premvos_amd import _lib, ops
    lib, st = _lib.load(), _lib.current_stream()
    x = ops.NHWC.alloc(1, 8, 8, 64)
    pk = ops.pack_conv(torch.randn((64, 64, 3, 3)), None)
    out = ops.NHWC.alloc(1, 8, 8, 64)
    d = ops.conv_desc(x, pk, out, pad=(1, 1), tile_hint=6)
    assert not ops.pwdma_applicable(d)
    assert lib.premvos_conv2d_f32(C.byref(d), st) != 0 and b"LDS-DMA" in lib.premvos_last_error()
