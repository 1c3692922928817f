"""GPU parity: every flow-path kernel through the C-ABI vs the CPU oracle / plain torch fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import pwc_oracle as O  # noqa: E402


def _ops():
    from premvos_amd import ops
    return ops


def _to_nhwc(t, ops, ps=None, coff=0):
    n, c, h, w = t.shape
    total = (coff + c + 3) // 4 * 4 if ps is None else ps
    buf = torch.zeros((n, h, w, total), dtype=torch.float32, device="cuda")
    buf[..., coff:coff + c] = t.permute(0, 2, 3, 1).cuda()
    return ops.NHWC(buf, c=c, coff=coff)


CONV_CASES = [
    # n, cin, h, w, cout, k, stride, dil, pad(t,l,b,r), act, res
    (2, 3, 64, 64, 16, 3, 2, 1, (1, 1, 1, 1), "leaky", False),
    (1, 16, 33, 47, 16, 3, 1, 1, (1, 1, 1, 1), "leaky", False),
    (1, 117, 32, 48, 128, 3, 1, 1, (1, 1, 1, 1), "leaky", False),
    (1, 565, 16, 24, 2, 3, 1, 1, (1, 1, 1, 1), "none", True),
    (1, 128, 20, 20, 96, 3, 1, 8, (8, 8, 8, 8), "leaky", False),
    (1, 96, 20, 36, 64, 3, 1, 16, (16, 16, 16, 16), "leaky", False),
    (1, 196, 8, 14, 196, 3, 1, 1, (1, 1, 1, 1), "leaky", False),
    (1, 3, 75, 133, 64, 7, 2, 1, (2, 2, 3, 3), "relu", False),     # resnet conv0, asymmetric pad
    (2, 64, 31, 29, 256, 1, 1, 1, (0, 0, 0, 0), "relu", True),     # bottleneck 1x1 + shortcut
    (1, 256, 30, 30, 128, 3, 2, 1, (0, 0, 1, 1), "relu", False),   # stride-2 3x3 with pad [0,1]
    (1, 4, 65, 65, 32, 3, 2, 1, (0, 0, 0, 0), "relu", False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_torch(case):
    ops = _ops()
    n, cin, h, w, cout, k, s, dil, pad, act, use_res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    pt, pl, pb, pr = pad
    xp = F.pad(x, (pl, pr, pt, pb))
    ref = F.conv2d(xp.double(), wt.double(), b.double(), stride=s, dilation=dil)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + res.double()
    if act == "leaky":
        ref = F.leaky_relu(ref, 0.1)
    elif act == "relu":
        ref = F.relu(ref)
    xin = _to_nhwc(x, ops, coff=32, ps=(32 + cin + 3) // 4 * 4 + 8)       # a window of a wider buffer
    out = _to_nhwc(torch.zeros(ref.shape), ops, coff=5, ps=cout + 9)
    pk = ops.pack_conv(wt, b)
    rr = _to_nhwc(res, ops) if res is not None else None
    ops.conv2d(xin, pk, out, stride=(s, s), dilation=(dil, dil), pad=(pt, pl),
               act={"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY}[act], res=rr)
    torch.cuda.synchronize()
    got = out.torch().cpu().double()
    err = (got - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, ref.abs().max().item()), err
    # channels outside the destination window stay untouched (zero)
    assert out.buf[..., :5].abs().max().item() == 0 and out.buf[..., 5 + cout:].abs().max().item() == 0


@pytest.mark.parametrize("hint", [(256, 128), (128, 128), (128, 96), (128, 64), (128, 32), (64, 128), (64, 64), (64, 32)])
def test_conv2d_every_tile_config(hint):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1, 40, 37, 41), generator=g)
    wt = torch.randn((100, 40, 3, 3), generator=g) * 0.07
    b = torch.randn((100,), generator=g)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    out = _to_nhwc(torch.zeros(ref.shape), ops)
    ops.conv2d(_to_nhwc(x, ops), ops.pack_conv(wt, b), out, pad=(1, 1), tile_hint=(hint[0] << 16) | hint[1])
    torch.cuda.synchronize()
    assert (out.torch().cpu().double() - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("cin,cout,h,w", [(2, 2, 8, 14), (37, 2, 16, 12), (661, 2, 8, 8)])
def test_deconv4x4s2_matches_torch(cin, cout, h, w):
    ops = _ops()
    g = torch.Generator().manual_seed(cin)
    x = torch.randn((2, cin, h, w), generator=g)
    wt = torch.randn((cin, cout, 4, 4), generator=g) * (1.0 / cin) ** 0.5
    b = torch.randn((cout,), generator=g)
    ref = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=2, padding=1)
    out = _to_nhwc(torch.zeros(ref.shape), ops, coff=3, ps=12)
    ops.conv2d(_to_nhwc(x, ops), ops.pack_deconv4x4s2(wt, b), out, pad=(1, 1))
    torch.cuda.synchronize()
    assert (out.torch().cpu().double() - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("c,h,w", [(196, 8, 14), (128, 16, 28), (32, 33, 50), (4, 5, 3)])
def test_corr_matches_oracle(c, h, w):
    ops = _ops()
    g = torch.Generator().manual_seed(c + h)
    f1 = torch.randn((2, c, h, w), generator=g)
    f2 = torch.randn((2, c, h, w), generator=g)
    ref = F.leaky_relu(torch.from_numpy(O.correlation_np(f1.numpy(), f2.numpy())), 0.1)
    out = ops.NHWC.alloc(2, h, w, 81 + c + 7)
    ops.corr(_to_nhwc(f1, ops), _to_nhwc(f2, ops), out.slice(4, 81 + c), 4, 0.1, True)
    torch.cuda.synchronize()
    got = out.slice(4, 81).torch().cpu()
    assert (got - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(out.slice(4 + 81, c).torch().cpu(), f1)
    assert out.buf[..., :4].abs().max().item() == 0


@pytest.mark.parametrize("c,h,w", [(64, 64, 112), (96, 19, 70), (20, 9, 33)])
def test_corr_tiled_multi_tile_shapes(c, h, w):
    """Several 8x32 tiles per image incl. ragged right / bottom tiles and a partial last 16-channel chunk; no copy of f1,
    no LeakyReLU (slope 1)."""
    ops = _ops()
    g = torch.Generator().manual_seed(c * h + w)
    f1 = torch.randn((3, c, h, w), generator=g)
    f2 = torch.randn((3, c, h, w), generator=g)
    ref = O.correlation_torch(f1, f2)
    out = ops.NHWC.alloc(3, h, w, 81 + 3)
    out.buf.fill_(7.0)
    ops.corr(_to_nhwc(f1, ops), _to_nhwc(f2, ops), out.slice(0, 81), 4, 1.0, False)
    torch.cuda.synchronize()
    got = out.slice(0, 81).torch().cpu()
    assert (got - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    assert torch.all(out.buf[..., 81:] == 7.0)          # nothing written beyond the 81 channels


def test_corr_instantiations_write_the_same_bits():
    """Round 6: the cost volume is a template over (tile, channels per LDS chunk, waves per SIMD, prefetch, pixels per thread) and the
    launcher picks an instantiation by shape (variant 5 for C <= 32 or small maps, 11 = two pixels per thread for large 64-channel
    maps, 0 = the round-2 form elsewhere).  PREMVOS_CORR_VARIANT forces one (read once per process: one subprocess each).  On ragged
    shapes -- partial tiles right and below, a partial last chunk, a destination window that is not 16-byte aligned, the copy of f1
    -- every instantiation must write the bits of the round-2 form (same products, same order), and those match the oracle."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from premvos_amd import ops
for c, h, w, off in ((32, 37, 70, 4), (64, 21, 45, 0), (20, 9, 33, 3), (96, 17, 40, 8)):
    g = torch.Generator().manual_seed(c + h)
    f1 = torch.randn((3, h, w, c + (-c) %% 4), generator=g).cuda()
    f2 = torch.randn((3, h, w, c + (-c) %% 4), generator=g).cuda()
    out = ops.NHWC.alloc(3, h, w, off + 81 + c + 5)
    out.buf.fill_(3.0)
    ops.corr(ops.NHWC(f1, c=c), ops.NHWC(f2, c=c), out.slice(off, 81 + c), 4, 0.1, True)
    torch.cuda.synchronize()
    print(int(out.buf.view(torch.int32).to(torch.int64).sum().item()), float(out.buf[..., off:off + 81].abs().max().item()))
""" % repo
    outs = {}
    for v in ("0", "2", "5", "9", "11", "13", ""):
        env = dict(os.environ, PYTHONPATH=repo)
        env.pop("PREMVOS_CORR_VARIANT", None)
        if v:
            env["PREMVOS_CORR_VARIANT"] = v
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[v] = [ln for ln in r.stdout.splitlines() if ln and ln[0] in "-0123456789"]
        assert len(outs[v]) == 4
    for v, lines in outs.items():
        assert lines == outs["0"], (v, lines, outs["0"])


def test_corr_unaligned_destination_window():
    """A channel window that does not start on a 16-byte boundary takes the 4-byte store path."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    f1 = torch.randn((2, 32, 17, 40), generator=g)
    f2 = torch.randn((2, 32, 17, 40), generator=g)
    ref = F.leaky_relu(O.correlation_torch(f1, f2), 0.1)
    out = ops.NHWC.alloc(2, 17, 40, 3 + 81 + 32)
    ops.corr(_to_nhwc(f1, ops), _to_nhwc(f2, ops), out.slice(3, 81 + 32), 4, 0.1, True)
    torch.cuda.synchronize()
    assert (out.slice(3, 81).torch().cpu() - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(out.slice(3 + 81, 32).torch().cpu(), f1) and out.buf[..., :3].abs().max().item() == 0


def test_corr_general_md_falls_back():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    f1 = torch.randn((1, 8, 12, 20), generator=g)
    f2 = torch.randn((1, 8, 12, 20), generator=g)
    ref = torch.from_numpy(O.correlation_np(f1.numpy(), f2.numpy(), 2, 1, 2, 1, 1))
    out = ops.NHWC.alloc(1, 12, 20, 25)
    ops.corr(_to_nhwc(f1, ops), _to_nhwc(f2, ops), out, 2, 1.0, False)
    torch.cuda.synchronize()
    assert (out.torch().cpu() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("c,h,w,mag", [(128, 16, 28, 3.0), (32, 40, 75, 6.0), (64, 8, 14, 1e9)])
def test_warp_corr_fused_is_bit_identical_to_warp_then_corr(c, h, w, mag):
    """premvos_warp_corr_fwd_f32 == premvos_warp_fwd_f32 -> premvos_corr_fwd_f32 (PWCNet.py:207-208), every bit."""
    ops = _ops()
    g = torch.Generator().manual_seed(c + w)
    f1 = torch.randn((2, c, h, w), generator=g)
    x2 = torch.randn((2, c, h, w), generator=g)
    flo = torch.randn((2, 2, h, w), generator=g) * mag
    flo[0, :, 0, 0] = 0.0
    a, x, f = _to_nhwc(f1, ops), _to_nhwc(x2, ops), _to_nhwc(flo, ops)
    wbuf = ops.NHWC.alloc(2, h, w, c)
    two = ops.NHWC.alloc(2, h, w, 81 + c)
    one = ops.NHWC.alloc(2, h, w, 81 + c)
    ops.warp(x, f, 1.25, wbuf)
    ops.corr(a, wbuf, two, 4, 0.1, True)
    ops.warp_corr(a, x, f, 1.25, one, 4, 0.1, True)
    torch.cuda.synchronize()
    assert torch.equal(one.buf, two.buf)
    ref = F.leaky_relu(O.correlation_torch(f1, O.warp(x2, flo * 1.25)), 0.1)
    bad = (one.slice(0, 81).torch().cpu() - ref).abs() > 1e-5 * (1 + ref.abs())
    assert bad.float().mean().item() < 1e-3          # samples at the 0.9999 mask threshold / a cell edge may flip


def test_corr_reference_known_answer():
    """test/test.py:76-77: correlation(0,1,0,1,1,1) on [[1,2],[3,4]] x [[5,6],[7,8]] = [[5,12],[21,32]];
    :81: correlation(1,1,1,1,1,1) gives 1x9x2x2."""
    ops = _ops()
    a = torch.tensor([[1., 2.], [3., 4.]]).view(1, 1, 2, 2).cuda()
    b = torch.tensor([[5., 6.], [7., 8.]]).view(1, 1, 2, 2).cuda()
    y = ops.corr_nchw(a, b, 0, 1, 0, 1, 1, 1)
    assert torch.equal(y.cpu().view(2, 2), torch.tensor([[5., 12.], [21., 32.]]))
    y2 = ops.corr_nchw(a, b, 1, 1, 1, 1, 1, 1)
    assert tuple(y2.shape) == (1, 9, 2, 2)
    assert torch.equal(y2.cpu(), torch.from_numpy(O.correlation_np(a.cpu().numpy(), b.cpu().numpy(), 1, 1, 1, 1, 1)))


@pytest.mark.parametrize("args", [(3, 3, 4, 1, 2), (1, 3, 2, 2, 1), (20, 1, 20, 1, 2), (4, 1, 4, 1, 1)])
def test_corr_nchw_general_geometry(args):
    ops = _ops()
    pad, ks, md, s1, s2 = args
    g = torch.Generator().manual_seed(sum(args))
    a = torch.randn((2, 5, 24, 30), generator=g)
    b = torch.randn((2, 5, 24, 30), generator=g)
    ref = O.correlation_np(a.numpy(), b.numpy(), pad, ks, md, s1, s2)
    got = ops.corr_nchw(a.cuda(), b.cuda(), pad, ks, md, s1, s2, 1).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 1e-5


@pytest.mark.parametrize("c,h,w,mag", [(128, 16, 28, 3.0), (32, 64, 48, 10.0), (4, 7, 9, 1.0), (64, 8, 8, 1e9)])
def test_warp_matches_oracle(c, h, w, mag):
    ops = _ops()
    g = torch.Generator().manual_seed(c * h)
    x = torch.randn((2, c, h, w), generator=g)
    flo = torch.randn((2, 2, h, w), generator=g) * mag
    flo[0, :, 0, 0] = 0.0                 # exact identity sample
    flo[0, :, 1, 1] = torch.tensor([1.0, -1.0])   # integer shift
    scale = 1.25
    ref = O.warp(x, flo * scale)
    out = ops.NHWC.alloc(2, h, w, c)
    ops.warp(_to_nhwc(x, ops), _to_nhwc(flo, ops), scale, out)
    torch.cuda.synchronize()
    got = out.torch().cpu()
    bad = (got - ref).abs() > 1e-5 * (1 + ref.abs())
    # a sample landing within float rounding of the 0.9999 mask threshold / a cell edge may flip
    assert bad.float().mean().item() < 1e-4, bad.float().mean().item()


def test_layout_roundtrip():
    ops = _ops()
    x = torch.randn((3, 37, 19, 45))
    v = ops.NHWC.alloc(3, 19, 45, 37)
    ops.nchw_to_nhwc(x.cuda(), v)
    back = ops.nhwc_to_nchw(v)
    torch.cuda.synchronize()
    assert torch.equal(back.cpu(), x)
    assert torch.equal(v.buf[..., :37].cpu(), x.permute(0, 2, 3, 1))
    assert v.buf[..., 37:].abs().max().item() == 0


@pytest.mark.parametrize("case", [(1, 661, 16, 28, 128, 3, 0), (2, 1024, 6, 9, 256, 1, 0), (1, 117, 8, 14, 6, 3, 5),
                                  (1, 597, 32, 56, 96, 3, 3)])
def test_conv2d_split_k_deterministic_and_exact(case):
    """Small-M / long-K layers are cut along K; result equals the unsplit kernel to fp32 round-off and is bitwise
    reproducible run to run (fixed-order slab reduction, no atomics)."""
    ops = _ops()
    n, cin, h, w, cout, k, force = case
    g = torch.Generator().manual_seed(cin)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn((cout,), generator=g)
    res = torch.randn((n, cout, h, w), generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), padding=k // 2) + res.double(), 0.1)
    xin, pk, rr = _to_nhwc(x, ops), ops.pack_conv(wt, b), _to_nhwc(res, ops)
    outs = []
    for sk in (force, force, -1):
        out = _to_nhwc(torch.zeros(ref.shape), ops, coff=3, ps=cout + 5)
        d = ops.conv_desc(xin, pk, out, pad=(k // 2, k // 2), act=ops.ACT_LEAKY, res=rr, split_k=sk)
        if sk >= 0:
            assert ops.workspace_bytes(d) > 0          # auto mode decides to split these shapes
        ops.conv2d(xin, pk, out, pad=(k // 2, k // 2), act=ops.ACT_LEAKY, res=rr, split_k=sk)
        torch.cuda.synchronize()
        outs.append(out.torch().cpu())
    assert torch.equal(outs[0], outs[1])
    assert (outs[0].double() - ref).abs().max().item() < 2e-4
    assert (outs[0] - outs[2]).abs().max().item() < 1e-4


def test_deconv_split_k_pixel_shuffle():
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    x = torch.randn((1, 661, 8, 14), generator=g)
    wt = torch.randn((661, 2, 4, 4), generator=g) * 0.05
    b = torch.randn((2,), generator=g)
    ref = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=2, padding=1)
    out = _to_nhwc(torch.zeros(ref.shape), ops, coff=1, ps=8)
    ops.conv2d(_to_nhwc(x, ops), ops.pack_deconv4x4s2(wt, b), out, pad=(1, 1), split_k=4)
    torch.cuda.synchronize()
    assert (out.torch().cpu().double() - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("case", [(3, 200, 25, 25, 200, 1, (128, 128), 32, 4, 3), (2, 96, 31, 40, 72, 3, (64, 128), 16, 9, 4),
                                  (1, 64, 40, 52, 40, 3, (128, 64), 16, 1, 2)])
def test_conv2d_tail_split(case):
    """Tail split: the last rows of output tiles run as k-slices + fixed-order reduce.  Rows of the main launch are
    bit-identical to the plain kernel, tail rows agree to fp32 round-off, the whole thing is reproducible, and a
    missing workspace is an error (not a silent fallback)."""
    from premvos_amd import _lib
    import ctypes as C
    ops = _ops()
    n, cin, h, w, cout, k, (bm, bn), st, tail, ts = case
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn((cout,), generator=g)
    res = torch.randn((n, cout, h, w), generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=k // 2) + res.double())
    xin, pk, rr = _to_nhwc(x, ops), ops.pack_conv(wt, b), _to_nhwc(res, ops)
    outs = []
    for use_tail in (True, True, False):
        out = _to_nhwc(torch.zeros(ref.shape), ops, coff=1, ps=cout + 3)
        d = ops.conv_desc(xin, pk, out, pad=(k // 2, k // 2), act=ops.ACT_RELU, res=rr, split_k=-1, stage_k=st,
                          tile_hint=(bm << 16) | bn)
        if use_tail:
            d.tail_m_tiles, d.tail_split_k = tail, ts
            assert ops.workspace_bytes(d) > 0
            assert _lib.load().premvos_conv2d_f32(C.byref(d), _lib.current_stream()) != 0     # no workspace -> error
            assert "workspace" in _lib.load().premvos_last_error().decode()
        ws = ops.assign_workspace([d])          # noqa: F841
        ops.run_desc(d)
        torch.cuda.synchronize()
        outs.append(out.torch().cpu())
    assert torch.equal(outs[0], outs[1])
    assert (outs[0].double() - ref).abs().max().item() < 2e-4
    m = n * h * w
    main_rows = (-(-m // bm) - tail) * bm
    flat = [o.permute(0, 2, 3, 1).reshape(m, cout) for o in outs]
    assert torch.equal(flat[0][:main_rows], flat[2][:main_rows])
    assert (flat[0][main_rows:] - flat[2][main_rows:]).abs().max().item() < 1e-4
    assert not torch.equal(flat[0][main_rows:], torch.zeros_like(flat[0][main_rows:]))


@pytest.mark.parametrize("case", [(2, 565, 20, 28, 2, 3, 1, 1, False), (1, 32, 17, 23, 2, 3, 1, 1, True), (3, 256, 9, 9, 2, 1, 1, 1, False),
                                  (1, 117, 8, 14, 1, 3, 2, 2, False), (1, 661, 16, 28, 2, 3, 1, 1, False)])
def test_conv2d_direct_kernel_for_one_or_two_output_channels(case):
    """cout <= 2 heads (predict_flow, dc_conv7 + residual, logits) run on the direct per-pixel kernel; same numbers as
    the MFMA kernel (fp32 round-off: another summation order) and as torch; bit-identical from run to run."""
    ops = _ops()
    n, cin, h, w, cout, k, stride, dil, with_res = case
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn((cout,), generator=g)
    pad = dil * (k // 2)
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=pad, dilation=dil)
    res = torch.randn(ref.shape, generator=g) if with_res else None
    if with_res:
        ref = ref + res.double()
    ref = F.leaky_relu(ref, 0.1)
    xin, pk = _to_nhwc(x, ops), ops.pack_conv(wt, b)
    rr = _to_nhwc(res, ops) if with_res else None
    outs = []
    for hint in (1, 1, 0, (64 << 16) | 32):
        out = _to_nhwc(torch.zeros(ref.shape), ops, coff=2, ps=8)
        d = ops.conv_desc(xin, pk, out, stride=(stride, stride), dilation=(dil, dil), pad=(pad, pad), act=ops.ACT_LEAKY, res=rr,
                          tile_hint=hint, split_k=-1)
        if hint in (0, 1):
            assert ops.workspace_bytes(d) == 0
        ops.run_desc(d)
        torch.cuda.synchronize()
        outs.append(out.torch().cpu())
        assert out.buf[..., :2].abs().max().item() == 0 and out.buf[..., 2 + cout:].abs().max().item() == 0
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert (outs[0].double() - ref).abs().max().item() < 2e-4
    assert (outs[0] - outs[3]).abs().max().item() < 1e-4


@pytest.mark.parametrize("seed", range(12))
def test_pointwise_conv_interior_fast_path_boundaries(seed):
    """Round 3: interior stages of interior tiles of 1x1 layers request their operands with plain (unpredicated) loads and the
    K loop keeps two fragment sets.  Random shapes around every boundary of that path -- K a multiple of the stage depth or not,
    cin_pad < / = / > (KT - 1) * KB, ragged last M tile, ragged / narrow last column tile, stride 2, k-slices (kt_begin > 0), tail
    split, every tile -- against the fp64 convolution, and bit-identical across tiles (same k order per output)."""
    ops = _ops()
    rng = np.random.default_rng(1000 + seed)
    cin = int(rng.choice([4, 12, 16, 20, 31, 32, 36, 64, 100, 128, 132, 260, 728]))
    cout = int(rng.choice([32, 48, 88, 96, 128, 130, 200, 256, 728]))
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(9, 40)), int(rng.integers(9, 40))
    stride = int(rng.choice([1, 1, 2]))
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn((cout,), generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), stride=stride))
    ho, wo = ref.shape[2:]
    xin, pk = _to_nhwc(x, ops), ops.pack_conv(wt, b)
    outs = {}
    for bm, bn in ((128, 128), (64, 64), (128, 64), (64, 128), (128, 32)):
        if bn == 32 and cout > 32:
            continue
        for sk, tail in ((-1, (0, 0)), (2, (0, 0)), (-1, (1, 2))):
            if sk > 0 and pk.k_pad < 64:
                continue
            out = ops.NHWC.alloc(n, ho, wo, cout)
            d = ops.conv_desc(xin, pk, out, stride=(stride, stride), act=ops.ACT_RELU, tile_hint=(bm << 16) | bn, stage_k=16, split_k=sk)
            d.tail_m_tiles, d.tail_split_k = tail
            ws = torch.empty(max(ops.workspace_bytes(d) // 4, 1), dtype=torch.float32, device="cuda")
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
            ops.run_desc(d)
            torch.cuda.synchronize()
            got = out.torch().cpu()
            assert (got.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), (bm, bn, sk, tail)
            outs[(bm, bn, sk, tail)] = got
    for key, got in outs.items():
        if key[2] == -1 and key[3] == (0, 0):
            assert torch.equal(got, outs[[k for k in outs if k[2] == -1 and k[3] == (0, 0)][0]]), key      # tiles: order-neutral


@pytest.mark.parametrize("cin,cout,n,h,w,res,act", [(64, 128, 2, 37, 41, False, "relu"), (64, 256, 1, 50, 50, True, "relu"),
                                                    (128, 128, 3, 19, 23, True, "none"), (128, 512, 1, 31, 17, True, "relu"),
                                                    (128, 256, 1, 5, 7, False, "leaky"), (64, 128, 1, 300, 301, True, "relu"),
                                                    # cout = 384: three column tiles do not divide 256 CUs (85 x 3 workgroups
                                                    # left rows of the third tile unwritten); many steps, and fewer steps than walkers
                                                    (64, 384, 1, 173, 174, True, "relu"), (128, 384, 1, 10, 10, False, "none")])
def test_streaming_pointwise_kernel_is_bit_identical_to_the_implicit_gemm(cin, cout, n, h, w, res, act):
    """tile_hint 5 (csrc/conv_stream_f32.hip: persistent workgroups, weights resident in LDS, no workgroup barrier in the loop) adds
    the products of an output in the implicit GEMM's order: the same bits, whatever M (ragged last step, fewer steps than
    workgroups), with / without residual, per activation; and it refuses layers it does not cover."""
    ops = _ops()
    from premvos_amd import _lib
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn((cout,), generator=g)
    r = torch.randn((n, cout, h, w), generator=g) if res else None
    a = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY}[act]
    xin, pk, rr = _to_nhwc(x, ops), ops.pack_conv(wt, b), (_to_nhwc(r, ops) if res else None)
    outs = []
    for hint in ((128 << 16) | 128, 5, (64 << 16) | 64):
        out = ops.NHWC.alloc(n, h, w, cout)
        d = ops.conv_desc(xin, pk, out, act=a, res=rr, tile_hint=hint, stage_k=16, split_k=-1)
        assert ops.stream_applicable(d)
        assert ops.numerics_key(d, (hint, 16, -1, 0, 0)) == ops.numerics_key(d, (5, 0, -1, 0, 0))
        ops.run_desc(d)
        torch.cuda.synchronize()
        outs.append(out.torch().cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = F.conv2d(x.double(), wt.double(), b.double())
    ref = ref + r.double() if res else ref
    ref = {"none": lambda v: v, "relu": F.relu, "leaky": lambda v: F.leaky_relu(v, 0.1)}[act](ref)
    assert (outs[1].double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    # not covered: stride 2, cout not a multiple of 128
    wt2 = torch.randn((96, cin, 1, 1), generator=g)
    d = ops.conv_desc(xin, ops.pack_conv(wt2, None), ops.NHWC.alloc(n, h, w, 96), tile_hint=5)
    assert not ops.stream_applicable(d)
    with pytest.raises(_lib.PremvosError, match="streaming pointwise"):
        ops.run_desc(d)


@pytest.mark.parametrize("n,cin,h,w,res", [(2, 597, 64, 112, False), (3, 117, 45, 53, True), (1, 32, 128, 224, True), (2, 565, 16, 28, False)])
def test_two_channel_heads_tiled_form_matches_the_per_pixel_form_and_fp64(n, cin, h, w, res):
    """PWC-Net's predict_flow / dc_conv7 heads (PWCNet.py:131, 267: 3x3, two outputs) on csrc/conv_smalln_f32.hip: maps of >= 100
    4x4 tiles take the tiled form (every loaded pixel feeds its nine outputs), smaller ones the per-pixel form (stage_k = 1 forces
    it).  Both against torch's fp64 convolution; the choice depends on the map, never on the batch -- image 0 of a batch of n has
    the bits of a batch of one (a ragged last chunk must not change a frame's numbers)."""
    from premvos_amd import ops
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((2, cin, 3, 3), generator=g) * (1.0 / (9 * cin)) ** 0.5
    b = torch.randn((2,), generator=g)
    r = torch.randn((n, 2, h, w), generator=g) if res else None
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    if res:
        ref = ref + r.double()

    def nhwc(t, ps=None):
        v = ops.NHWC.alloc(t.shape[0], t.shape[2], t.shape[3], ps or t.shape[1])
        v.buf.zero_()
        v.buf[..., :t.shape[1]] = t.permute(0, 2, 3, 1).cuda()
        return v.slice(0, t.shape[1])

    pk = ops.pack_conv(wt, b)
    xin, rin = nhwc(x), (nhwc(r, 4) if res else None)
    got = {}
    for sk in (0, 1):
        out = ops.NHWC.alloc(n, h, w, 2)
        out.buf.fill_(5.0)
        ops.conv2d(xin, pk, out, pad=(1, 1), res=rin, tile_hint=1, stage_k=sk)
        torch.cuda.synchronize()
        got[sk] = out.torch().cpu()
        assert (got[sk].double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), sk
    tiled = ((h + 3) // 4) * ((w + 3) // 4) >= 100
    assert torch.equal(got[0], got[1]) == (not tiled)              # below the threshold both calls ARE the per-pixel form
    one = ops.NHWC.alloc(1, h, w, 2)
    ops.conv2d(nhwc(x[:1]), pk, one, pad=(1, 1), res=nhwc(r[:1], 4) if res else None, tile_hint=1)
    torch.cuda.synchronize()
    assert torch.equal(one.torch().cpu()[0], got[0][0])
