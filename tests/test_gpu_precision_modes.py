"""The bf16-MFMA modes of the dense conv (PREMVOS_PREC_BF16 / BF16X3): configs[2]/[4] of BASELINE.json ask for bf16
compute.  fp32 stays the parity/default mode; these tests state (and bound) what the two faster modes cost in accuracy."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import proposal_oracle as PO  # noqa: E402
from oracle import pwc_oracle as O  # noqa: E402
from oracle import refinement_oracle as RO  # noqa: E402

CASES = [(2, 3, 40, 44, 16, 3, 2, 1), (1, 117, 32, 48, 128, 3, 1, 1), (1, 565, 16, 24, 2, 3, 1, 1),
         (1, 128, 20, 20, 96, 3, 1, 8), (2, 64, 31, 29, 256, 1, 1, 1), (1, 728, 25, 25, 728, 1, 1, 1),
         (1, 661, 8, 14, 128, 3, 1, 1), (1, 256, 30, 30, 128, 3, 2, 1)]


@pytest.mark.parametrize("prec,tol", [("bf16x3", 3e-5), ("bf16", 1.5e-2)])
@pytest.mark.parametrize("case", CASES)
def test_conv_precision_modes(case, prec, tol):
    from premvos_amd import ops
    n, cin, h, w, cout, k, s, dil = case
    g = torch.Generator().manual_seed(cin * k + cout)
    x = torch.randn((n, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    pad = dil * (k // 2)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), b.double(), stride=s, dilation=dil, padding=pad), 0.1)
    xin = ops.NHWC.alloc(n, h, w, cin)
    xin.buf[..., :cin] = x.permute(0, 2, 3, 1).cuda()
    out = ops.NHWC.alloc(n, ref.shape[2], ref.shape[3], cout)
    pk = ops.pack_conv(wt, b, precision=prec)
    ops.conv2d(xin, pk, out, stride=(s, s), dilation=(dil, dil), pad=(pad, pad), act=ops.ACT_LEAKY)
    torch.cuda.synchronize()
    err = (out.torch().cpu().double() - ref).abs().max().item()
    assert err < tol * max(1.0, ref.abs().max().item()), err


def test_deconv_and_splitk_in_bf16x3():
    from premvos_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn((1, 661, 8, 14), generator=g)
    wt = torch.randn((661, 2, 4, 4), generator=g) * 0.05
    b = torch.randn((2,), generator=g)
    ref = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=2, padding=1)
    xin = ops.NHWC.alloc(1, 8, 14, 661)
    xin.buf[..., :661] = x.permute(0, 2, 3, 1).cuda()
    for sk in (-1, 4):
        out = ops.NHWC.alloc(1, 16, 28, 2)
        ops.conv2d(xin, ops.pack_deconv4x4s2(wt, b, precision="bf16x3"), out, pad=(1, 1), split_k=sk)
        torch.cuda.synchronize()
        assert (out.torch().cpu().double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("prec,tol", [("bf16x3", 1e-3), ("bf16", 0.5)])
def test_pwc_net_precision_modes(prec, tol):
    from premvos_amd.flow import pwc_dc_net
    sd = O.synth_state_dict(1)
    x = O.synth_frame_pair(128, 192, seed=12, shift=(1.5, -0.75))
    with torch.no_grad():
        ref = O.pwc_forward(sd, x)
    net = pwc_dc_net(None, precision=prec)
    net.load_state_dict(sd)
    got = net(x.cuda()).cpu()
    err = (got - ref).abs().max().item()
    print(f"PWC-Net {prec}: max |flow err| = {err:.3e} (|flow| max {ref.abs().max().item():.2f})")
    assert err < tol * max(1.0, ref.abs().max().item())


def test_proposal_and_refinement_bf16x3_stay_within_fp32_tolerances():
    """bf16x3 is accurate enough for the same 1e-3 bars as fp32 on the reduced-depth nets."""
    from premvos_amd.proposal import OfflinePredictor, ProposalNet
    from premvos_amd.refinement import RefinementNet
    w = PO.synth_weights(1, (2, 2, 3, 2))
    img = np.random.default_rng(1).integers(0, 256, (160, 256, 3), dtype=np.uint8)
    (fb, fp, fl, fi), inter = PO.model_forward(w, img, (2, 2, 3, 2), intermediates=True)
    net = ProposalNet(w, (2, 2, 3, 2), precision="bf16x3")
    OfflinePredictor(net)(img)
    p = net.plan(1, 160, 256)
    fm = p.featuremap.torch().cpu()
    e1 = (fm - inter["featuremap"]).abs().max().item() / max(1.0, inter["featuremap"].abs().max().item())
    n = int(p.roi_count.item())
    same = n == len(inter["proposal_idx"]) and np.array_equal(p.roi_idx[0, :n].cpu().numpy(), inter["proposal_idx"].astype(np.int32))
    print(f"proposal bf16x3: featuremap rel err {e1:.2e}; RPN indices identical: {same}")
    assert e1 < 1e-3
    rw = RO.synth_weights(1, 2)
    img2 = (np.random.default_rng(1).random((120, 200, 3)) * 255).astype(np.uint8)
    box = [20.4, 30.5, 90.6, 150.5]
    rnet = RefinementNet(rw, 2, precision="bf16x3")
    rp = rnet.refine(torch.from_numpy(img2).cuda(), torch.tensor([box]).cuda(), max_boxes=2, with_posterior=True)
    x, crop = RO.make_input(img2, box)
    with torch.no_grad():
        lg = RO.deeplab_logits(rw, x, 2)
    e2 = (rp.logits.torch().cpu()[:1] - lg).abs().max().item() / max(1.0, lg.abs().max().item())
    rm, rpost = RO.output_layer(lg, crop, 120, 200)
    e3 = np.abs(rp.posterior[0].cpu().numpy() - rpost).max()
    print(f"refinement bf16x3: logits rel err {e2:.2e}, posterior abs err {e3:.2e}")
    assert e2 < 1e-3 and e3 < 1e-3


@pytest.mark.parametrize("n,h,w,cin,cout,act,res", [(2, 25, 25, 728, 728, "none", True), (1, 49, 49, 256, 728, "relu", False),
                                                    (3, 13, 17, 128, 136, "leaky", True), (1, 9, 9, 2048, 256, "relu", False),
                                                    (1, 31, 29, 304, 256, "relu", False)])
def test_s8_depthwise_and_pointwise_bf16x3(n, h, w, cin, cout, act, res):
    """Round 4: premvos_dwconv3x3_f32 with PREMVOS_ACT_SPLIT8_BF16 stores the resident split layout S8 ({hi8, lo8} per group of 8
    channels) in place of its floats, and premvos_conv_bf16x3_s8_f32 multiplies it (hi.hi + hi.lo + lo.hi) -- against the fp64
    convolution of the fp32 depthwise result: fp32-class (3e-5 of the output scale), with ragged M / N tiles, residual, bias,
    activation, and an output that is a channel window of a wider buffer."""
    from premvos_amd import _lib, ops
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn((n, cin, h, w), generator=g)
    dw = torch.randn((cin, 1, 3, 3), generator=g) * (2.0 / 9) ** 0.5
    wt = torch.randn((cout, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    r = torch.randn((n, cout, h, w), generator=g) if res else None
    lib = _lib.load()
    xin = ops.NHWC.alloc(n, h, w, cin)
    xin.buf[..., :cin] = x.permute(0, 2, 3, 1).cuda()
    cpad = (cin + 3) // 4 * 4
    dwk = torch.zeros((9, cpad), device="cuda")
    dwk[:, :cin] = dw.view(cin, 9).t().cuda()
    bias0 = torch.zeros((cpad,), device="cuda")
    plain, split = ops.NHWC.alloc(n, h, w, cin), ops.NHWC.alloc_s8(n, h, w, cin)
    for t, flags in ((plain, 0), (split, _lib.ACT_SPLIT8_BF16)):
        _lib.check(lib.premvos_dwconv3x3_f32(xin.ptr, xin.ps, n, h, w, cin, dwk.data_ptr(), bias0.data_ptr(), cpad, t.ptr, t.ps, h, w,
                                             1, 1, 1, 1, 0, flags, _lib.current_stream()), "dw")
    torch.cuda.synchronize()
    # decode the S8 form: every 32 bytes = eight bf16 hi, eight bf16 lo
    raw = split.buf.view(torch.bfloat16).view(n, h, w, -1, 2, 8).float()
    hi, lo = raw[..., 0, :].reshape(n, h, w, -1)[..., :cin], raw[..., 1, :].reshape(n, h, w, -1)[..., :cin]
    ref_dw = plain.buf[..., :cin]
    assert torch.equal(hi, ref_dw.to(torch.bfloat16).float())                          # hi = round-to-nearest-even bf16
    assert ((hi + lo) - ref_dw).abs().max().item() <= 2.0 ** -15 * ref_dw.abs().max().item()
    assert torch.equal(split.torch(), (hi + lo).permute(0, 3, 1, 2))                   # (NHWC.torch() decodes an S8 buffer)
    pk = ops.pack_conv_s8(wt, b)
    out = ops.NHWC.alloc(n, h, w, cout + 8)
    out.buf.fill_(5.0)
    rin = None
    if res:
        rin = ops.NHWC.alloc(n, h, w, cout)
        rin.buf[..., :cout] = r.permute(0, 2, 3, 1).cuda()
    a = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY}[act]
    ops.conv_s8(split, pk, out.slice(0, cout), None, act=a, slope=0.1, res=rin)
    torch.cuda.synchronize()
    ref = F.conv2d(plain.torch().cpu().double(), wt.double(), b.double())
    if res:
        ref = ref + r.double()
    ref = {"none": lambda v: v, "relu": F.relu, "leaky": lambda v: F.leaky_relu(v, 0.1)}[act](ref)
    got = out.slice(0, cout).torch().cpu().double()
    assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    assert torch.all(out.buf[..., cout:] == 5.0)                                       # the channel window is respected
    # the layout flags are checked: an S8 output needs a pixel stride of whole groups
    bad = ops.NHWC.alloc(n, h, w, 4)
    assert lib.premvos_dwconv3x3_f32(xin.ptr, xin.ps, n, h, w, 4, dwk.data_ptr(), bias0.data_ptr(), 4, bad.ptr, bad.ps, h, w, 1, 1, 1, 1, 0,
                                     _lib.ACT_SPLIT8_BF16, _lib.current_stream()) != 0
    assert lib.premvos_dwconv3x3_f32(xin.ptr, xin.ps, n, h, w, cin, dwk.data_ptr(), bias0.data_ptr(), cpad, plain.ptr, plain.ps, h, w, 1, 1, 1,
                                     1, 0, 0x100, _lib.current_stream()) != 0          # round 3's flag is gone


def test_proposal_bf16x3_plan_uses_the_s8_chains(monkeypatch):
    """The bf16x3 proposal plan: from group 1 on the bottleneck chains live in the S8 layout (conv1 / conv2 / conv3 on
    csrc/conv_bf16x3_s8.hip, S8 residuals, no fp32 copy of the block outputs between a group's first and last block); group 0, the
    RPN 3x3 and the heads on the fp32 kernels; same feature map as the fp32 net on a reduced-depth net."""
    from premvos_amd import synth
    from premvos_amd.proposal.model import ProposalNet
    wts = synth.proposal_weights(0, num_blocks=(1, 2, 3, 2))
    img = synth.clip_frames(0, 1, 256, 384).cuda()
    nets = {p: ProposalNet(wts, num_blocks=(1, 2, 3, 2), precision=p, use_graph=False) for p in ("fp32", "bf16x3")}
    plans = {p: nets[p].run_resized(img) for p in nets}
    torch.cuda.synchronize()
    assert plans["bf16x3"].split_layers == 3 * (2 + 3 + 2) + 3 and plans["fp32"].split_layers == 0       # conv1-3 of groups 1-3 + 3 shortcuts
    fa, fb = plans["fp32"].featuremap.torch(), plans["bf16x3"].featuremap.torch()
    assert (fa - fb).abs().max().item() < 1e-3 * fa.abs().max().item()


@pytest.mark.parametrize("cin,stride,rate", [(20, 1, 1), (728, 1, 1), (64, 2, 1), (2048, 1, 6), (12, 1, 2)])
def test_s8_depthwise_store_paths(cin, stride, rate):
    """The S8 output of the three depthwise kernels (tile / row / per-pixel): with an even number of 4-channel units per pixel the
    two lanes of a group swap a half by DPP and store 16 bytes each, with an odd number every lane stores its two 8-byte halves;
    both decode to exactly bf16 hi + bf16 lo of the fp32 result."""
    from premvos_amd import _lib, ops
    g = torch.Generator().manual_seed(cin)
    n, h, w = 2, 26, 31
    ho, wo = (h + 1) // stride if stride == 2 else h, (w + 1) // stride if stride == 2 else w
    x = ops.NHWC.alloc(n, h, w, cin)
    x.buf[..., :cin] = torch.randn((n, h, w, cin), generator=g).cuda()
    cpad = (cin + 3) // 4 * 4
    dwk = torch.zeros((9, cpad), device="cuda")
    dwk[:, :cin] = (torch.randn((9, cin), generator=g) * 0.3).cuda()
    b0 = (torch.randn((cpad,), generator=g) * 0.1).cuda()
    lib = _lib.load()
    plain, s8 = ops.NHWC.alloc(n, ho, wo, cin), ops.NHWC.alloc_s8(n, ho, wo, cin)
    for t, fl in ((plain, ops.ACT_RELU), (s8, ops.ACT_RELU | _lib.ACT_SPLIT8_BF16)):
        _lib.check(lib.premvos_dwconv3x3_f32(x.ptr, x.ps, n, h, w, cin, dwk.data_ptr(), b0.data_ptr(), cpad, t.ptr, t.ps, ho, wo, stride,
                                             rate, rate if stride == 1 else 1, rate if stride == 1 else 1, 1, fl, _lib.current_stream()), "dw")
    torch.cuda.synchronize()
    ref = plain.buf[..., :cin]
    hi = ref.to(torch.bfloat16).float()
    lo = (ref - hi).to(torch.bfloat16).float()
    assert torch.equal(s8.torch(), (hi + lo).permute(0, 3, 1, 2))
