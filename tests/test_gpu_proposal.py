"""GPU parity of the proposal_net path vs the CPU oracle (oracle/proposal_oracle.py).
Integer/index results (top-k set, NMS selections, counts) must be bit-exact; float tensors within 1e-3 relative
(fp32 conv sums are ordered differently from the CPU BLAS)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import proposal_oracle as P  # noqa: E402

SMALL = (2, 2, 3, 2)


def _ops():
    from premvos_amd import _lib, ops
    return _lib, ops


def test_maxpool_padded():
    _lib, ops = _ops()
    x = torch.randn((2, 8, 11, 13))
    ref = F.max_pool2d(F.pad(x, (0, 1, 0, 1)), 3, stride=2)
    xin = ops.NHWC.alloc(2, 11, 13, 8)
    xin.buf[..., :8] = x.permute(0, 2, 3, 1).cuda()
    out = ops.NHWC.alloc(2, ref.shape[2], ref.shape[3], 8)
    _lib.check(_lib.load().premvos_maxpool_f32(xin.ptr, xin.ps, 2, 11, 13, 8, out.ptr, out.ps, out.h, out.w, 3, 2, 0,
                                               0, 0.0, _lib.current_stream()))
    assert torch.equal(out.torch().cpu(), ref)


def _rpn_case(fh, fw, seed, ties=False, img_hw=None):
    g = torch.Generator().manual_seed(seed)
    lab = torch.randn((fh, fw, 15), generator=g) * 3
    if ties:
        lab = torch.round(lab)            # many equal logits: exercises the tie rule at the top-k boundary
    box = torch.randn((fh, fw, 15, 4), generator=g) * 0.5
    box[..., 2:] *= 2
    h, w = img_hw if img_hw else (fh * 16 + 5, fw * 16 + 13)
    dec = P.decode_bbox_target(box.numpy(), P.all_anchors(fh, fw)).reshape(-1, 4)
    pb, ps, pidx = P.generate_rpn_proposals(dec, lab.numpy().reshape(-1), h, w)
    return lab, box, h, w, pb, ps, pidx


@pytest.mark.parametrize("fh,fw,seed,ties", [(10, 16, 0, False), (46, 83, 1, False), (5, 7, 2, False),
                                             (12, 20, 3, True), (46, 83, 4, True), (3, 4, 5, False)])
def test_rpn_proposals_bit_exact_indices(fh, fw, seed, ties):
    _lib, ops = _ops()
    from premvos_amd.proposal import cell_anchors
    lab, box, h, w, pb, ps, pidx = _rpn_case(fh, fw, seed, ties)
    assert np.array_equal(cell_anchors(), P.cell_anchors())
    rpn = ops.NHWC.alloc(1, fh, fw, 75)
    rpn.buf[0, :, :, :15] = lab.cuda()
    rpn.buf[0, :, :, 15:75] = box.reshape(fh, fw, 60).cuda()
    ca = torch.from_numpy(cell_anchors()).cuda()
    ob = torch.zeros((1, 100, 4), device="cuda")
    osc = torch.zeros((1, 100), device="cuda")
    oi = torch.zeros((1, 100), dtype=torch.int32, device="cuda")
    oc = torch.zeros((1,), dtype=torch.int32, device="cuda")
    _lib.check(_lib.load().premvos_rpn_proposals_f32(
        rpn.ptr, rpn.ps, 1, fh, fw, 15, 0, 15, ca.data_ptr(), 16.0, float(h), float(w), 1000, 100, 0.7, 0.0,
        float(P.BBOX_DECODE_CLIP), ob.data_ptr(), osc.data_ptr(), oi.data_ptr(), oc.data_ptr(),
        _lib.current_stream()))
    n = int(oc.item())
    assert n == len(pidx)
    assert np.array_equal(oi[0, :n].cpu().numpy(), pidx.astype(np.int32))           # bit-exact indices, in order
    assert np.array_equal(osc[0, :n].cpu().numpy(), ps)
    assert np.abs(ob[0, :n].cpu().numpy() - pb).max() < 1e-3                         # expf vs np.exp ulps
    assert (oi[0, n:].cpu().numpy() == -1).all()


@pytest.mark.parametrize("H,W,c,seed", [(10, 16, 8, 0), (46, 83, 16, 1)])
def test_roi_align_matches_oracle(H, W, c, seed):
    _lib, ops = _ops()
    g = torch.Generator().manual_seed(seed)
    fm = torch.randn((1, c, H, W), generator=g)
    n = 12
    xy = torch.rand((n, 2), generator=g) * torch.tensor([W * 16.0, H * 16.0])
    wh = torch.rand((n, 2), generator=g) * torch.tensor([W * 10.0, H * 10.0]) + 1
    boxes = torch.cat([xy - wh / 2, xy + wh / 2], 1).numpy().astype(np.float32)
    boxes[0] = [0, 0, W * 16, H * 16]
    boxes[1] = [-50, -40, 30, 20]                      # partly outside: extrapolation 0
    ref = P.roi_align(fm, boxes * np.float32(1.0 / 16), 14)
    f = ops.NHWC.alloc(1, H, W, c)
    f.buf[0, :, :, :c] = fm[0].permute(1, 2, 0).cuda()
    rois = torch.zeros((1, 16, 4), device="cuda")
    rois[0, :n] = torch.from_numpy(boxes).cuda()
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    out = ops.NHWC.alloc(16, 14, 14, c)
    out.buf.fill_(7.0)
    _lib.check(_lib.load().premvos_roi_align_f32(f.ptr, f.ps, 1, H, W, c, rois.data_ptr(), cnt.data_ptr(), 16,
                                                 1.0 / 16, 14, out.ptr, out.ps, _lib.current_stream()))
    got = out.torch().cpu()
    assert (got[:n] - ref).abs().max().item() < 1e-5
    assert got[n:].abs().max().item() == 0


def test_frcnn_tail_matches_oracle():
    _lib, ops = _ops()
    g = torch.Generator().manual_seed(9)
    n, h, w = 100, 300, 500
    xy = torch.rand((n, 2), generator=g) * torch.tensor([w * 1.0, h * 1.0])
    wh = torch.rand((n, 2), generator=g) * 150 + 4
    props = torch.cat([xy - wh / 2, xy + wh / 2], 1).clamp(min=0).numpy().astype(np.float32)
    cls = (torch.randn((n, 2), generator=g) * 2).numpy()
    cls[5] = cls[6]                                    # equal scores
    box = (torch.randn((n, 1, 4), generator=g)).numpy()
    fb, fp, fl, fi = P.fastrcnn_tail(cls, box, props, h, w)
    head = torch.zeros((n, 88), device="cuda")
    head[:, :2] = torch.from_numpy(cls).cuda()
    head[:, 2:6] = torch.from_numpy(box[:, 0]).cuda()
    rois = torch.from_numpy(props).cuda().view(1, n, 4).contiguous()
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    ob = torch.zeros((1, 20, 4), device="cuda")
    op = torch.zeros((1, 20), device="cuda")
    oi = torch.zeros((1, 20), dtype=torch.int32, device="cuda")
    oc = torch.zeros((1,), dtype=torch.int32, device="cuda")
    _lib.check(_lib.load().premvos_frcnn_tail_f32(head.data_ptr(), 88, rois.data_ptr(), cnt.data_ptr(), 1, n,
                                                  float(h), float(w), 0.5, 0.5, 20, float(P.BBOX_DECODE_CLIP), 10.0,
                                                  10.0, 5.0, 5.0, ob.data_ptr(), op.data_ptr(), oi.data_ptr(),
                                                  oc.data_ptr(), _lib.current_stream()))
    m = int(oc.item())
    assert m == len(fi) and m > 3
    assert np.array_equal(oi[0, :m].cpu().numpy(), fi.astype(np.int32))
    assert np.abs(op[0, :m].cpu().numpy() - fp).max() < 1e-6
    assert np.abs(ob[0, :m].cpu().numpy() - fb).max() < 1e-3


def test_proposal_preprocess_bit_exact():
    from oracle import cv_resize_oracle as R
    _lib, ops = _ops()
    img = np.random.default_rng(0).integers(0, 256, (60, 107, 3), dtype=np.uint8)
    nh, nw = P.custom_resize_shape(60, 107)
    assert (nh, nw) == (747, 1333)
    nh, nw = 94, 167
    ref = P.image_preprocess(torch.from_numpy(R.resize_linear_u8(img, nw, nh)))[0].permute(1, 2, 0)
    out = ops.NHWC.alloc(1, nh, nw, 3)
    _lib.check(_lib.load().premvos_proposal_preprocess_u8(torch.from_numpy(img).cuda().data_ptr(), 1, 60, 107,
                                                          out.ptr, nh, nw, 0, _lib.current_stream()))
    got = out.buf[0, :, :, :3].cpu()
    assert (got - ref).abs().max().item() < 1e-6
    rgb = np.ascontiguousarray(img[:, :, ::-1])
    _lib.check(_lib.load().premvos_proposal_preprocess_u8(torch.from_numpy(rgb).cuda().data_ptr(), 1, 60, 107,
                                                          out.ptr, nh, nw, 1, _lib.current_stream()))
    assert (out.buf[0, :, :, :3].cpu() - ref).abs().max().item() < 1e-6


@pytest.mark.parametrize("use_graph", [False, True])
def test_proposal_net_end_to_end_small(use_graph):
    """Reduced depth (2,2,3,2) at 160x256: every stage compared, indices bit-exact."""
    from premvos_amd.proposal import OfflinePredictor, ProposalNet
    w = P.synth_weights(1, SMALL)
    img = np.random.default_rng(1).integers(0, 256, (160, 256, 3), dtype=np.uint8)
    (fb, fp, fl, fi), inter = P.model_forward(w, img, SMALL, intermediates=True)
    net = ProposalNet(w, SMALL, use_graph=use_graph)
    out = OfflinePredictor(net)(img)
    p = net.plan(1, 160, 256)
    fm = p.featuremap.torch().cpu()
    ref_fm = inter["featuremap"]
    assert (fm - ref_fm).abs().max().item() < 1e-3 * max(1.0, ref_fm.abs().max().item())
    rl = p.rpn_out.buf[0, :, :, :15].cpu()
    assert (rl - inter["rpn_logits"]).abs().max().item() < 2e-3 * max(1.0, inter["rpn_logits"].abs().max().item())
    n = int(p.roi_count.item())
    assert n == len(inter["proposal_idx"])
    assert np.array_equal(p.roi_idx[0, :n].cpu().numpy(), inter["proposal_idx"].astype(np.int32))
    assert np.abs(p.rois[0, :n].cpu().numpy() - inter["proposals"]).max() < 0.05
    f5 = p.feat5.torch().cpu()[:n]
    assert (f5 - inter["feat5"]).abs().max().item() < 2e-3 * max(1.0, inter["feat5"].abs().max().item())
    boxes, probs, labels = out[0], out[1], out[2]
    assert len(boxes) == len(fb) and np.array_equal(net.outputs(p, 0)[6], fi)
    assert np.abs(probs - fp).max() < 1e-3 and np.abs(boxes - fb).max() < 0.05
    assert len(out) == 6 and out[3].shape == (len(fb), 2) and out[5].shape == (len(fb), 81)


def test_detect_one_image_and_json(tmp_path):
    from premvos_amd.proposal import OfflinePredictor, ProposalNet, convert_results_to_json, detect_one_image
    w = P.synth_weights(2, SMALL)
    img = np.random.default_rng(2).integers(0, 256, (60, 107, 3), dtype=np.uint8)
    rb, rp, _ = P.detect_one_image(w, img, SMALL)
    ref = P.results_to_json(rb, rp)
    res = detect_one_image(img, OfflinePredictor(ProposalNet(w, SMALL)))
    js = convert_results_to_json(res)
    assert len(js) == len(ref) and len(js) > 0
    for a, b in zip(js, ref):
        assert abs(a["score"] - b["score"]) <= 0.011
        assert np.abs(np.array(a["bbox"]) - np.array(b["bbox"])).max() <= 0.11
        assert all(isinstance(v, float) for v in a["bbox"])


def test_mask_head_mode_mask():
    """the mask branch (only with MODE_MASK; --forward switches it off, train.py:636-637)."""
    from premvos_amd import rle
    from premvos_amd.proposal import OfflinePredictor, ProposalNet, convert_results_to_json, detect_one_image
    w = P.synth_weights(4, SMALL)
    img = np.random.default_rng(4).integers(0, 256, (160, 256, 3), dtype=np.uint8)
    (fb, fp, fl, fi), inter = P.model_forward(w, img, SMALL, intermediates=True)
    ref_masks = P.maskrcnn_masks(w, inter["featuremap"], fb, SMALL[3])
    net = ProposalNet(w, SMALL, mode_mask=True)
    out = OfflinePredictor(net)(img)
    assert len(out) == 7 and out[6].shape == ref_masks.shape == (len(fb), 14, 14)
    assert np.abs(out[6] - ref_masks).max() < 2e-3
    # host paste (eval.py:35-58) + JSON with RLE (train.py:419-425)
    res = detect_one_image(img, OfflinePredictor(net))
    js = convert_results_to_json(res)
    assert all("segmentation" in j and j["segmentation"]["size"] == [160, 256] for j in js)
    assert len(js) == len(res) > 0 and res[0].mask.shape == (160, 256)
    assert np.array_equal(rle.decode(js[0]["segmentation"]), res[0].mask)
    # the host paste itself, on identical inputs, equals the oracle's restatement of eval.py:35-58
    from premvos_amd.proposal.driver import fill_full_mask
    for box, m in zip(fb.astype(np.float64), ref_masks):
        assert np.array_equal(fill_full_mask(box, m, (160, 256)), P.fill_full_mask(box, m, (160, 256)))


from hypothesis import given, settings, strategies as st  # noqa: E402


@settings(max_examples=20, deadline=None, derandomize=True)
@given(st.integers(0, 2 ** 31), st.integers(2, 14), st.integers(2, 20))
def test_rpn_proposals_kernel_fuzz_with_exact_ties_and_duplicate_boxes(seed, fh, fw):
    """Property test of the fused top-k + decode + clip + NMS kernel against the oracle on inputs built to collide: logits from a
    five-value set (ties everywhere, also across the top-k boundary) and box deltas from a three-value set (many anchors decode
    to bit-identical boxes, i.e. IoU exactly 1 and IoUs that recur).  Indices, order, scores and count must agree exactly."""
    _lib, ops = _ops()
    from premvos_amd.proposal import cell_anchors
    rng = np.random.default_rng(seed)
    lab = torch.from_numpy(rng.choice(np.array([-2.0, 0.0, 0.5, 0.5, 3.0], np.float32), (fh, fw, 15)))
    box = torch.from_numpy(rng.choice(np.array([-0.25, 0.0, 0.5], np.float32), (fh, fw, 15, 4)))
    h, w = fh * 16 + int(rng.integers(0, 16)), fw * 16 + int(rng.integers(0, 16))
    dec = P.decode_bbox_target(box.numpy(), P.all_anchors(fh, fw)).reshape(-1, 4)
    pb, ps, pidx = P.generate_rpn_proposals(dec, lab.numpy().reshape(-1), h, w)
    rpn = ops.NHWC.alloc(1, fh, fw, 75)
    rpn.buf[0, :, :, :15] = lab.cuda()
    rpn.buf[0, :, :, 15:75] = box.reshape(fh, fw, 60).cuda()
    ca = torch.from_numpy(cell_anchors()).cuda()
    ob = torch.zeros((1, 100, 4), device="cuda")
    osc = torch.zeros((1, 100), device="cuda")
    oi = torch.zeros((1, 100), dtype=torch.int32, device="cuda")
    oc = torch.zeros((1,), dtype=torch.int32, device="cuda")
    _lib.check(_lib.load().premvos_rpn_proposals_f32(
        rpn.ptr, rpn.ps, 1, fh, fw, 15, 0, 15, ca.data_ptr(), 16.0, float(h), float(w), 1000, 100, 0.7, 0.0,
        float(P.BBOX_DECODE_CLIP), ob.data_ptr(), osc.data_ptr(), oi.data_ptr(), oc.data_ptr(), _lib.current_stream()))
    n = int(oc.item())
    assert n == len(pidx), (n, len(pidx))
    assert np.array_equal(oi[0, :n].cpu().numpy(), pidx.astype(np.int32))
    assert np.array_equal(osc[0, :n].cpu().numpy(), ps)


def test_roi_align_row_kernel_equals_the_per_bin_kernel_on_adversarial_boxes(tmp_path):
    """Round 5: `premvos_roi_align_f32` runs the row-walking kernel (a thread keeps the corner pixels of its four feature-map rows in
    registers); PREMVOS_ROI_ALIGN=bin selects the per-bin kernel it replaced.  Same bits on boxes that leave the map on every side,
    degenerate (zero / negative extent), sub-pixel, integer-aligned (floor == ceil) and frame-sized boxes, ragged counts."""
    import os
    import subprocess
    import sys
    script = tmp_path / "roi.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})\n"
        "from premvos_amd import _lib, ops\n"
        "lib = _lib.load(); torch.manual_seed(3)\n"
        "B, R, fh, fw, C = 3, 24, 23, 37, 72\n"
        "fm = ops.NHWC(torch.randn((B, fh, fw, C), device='cuda'), c=C)\n"
        "W, H = fw * 16.0, fh * 16.0\n"
        "rng = np.random.default_rng(11)\n"
        "special = [[-50, -40, 80, 90], [W - 60, H - 30, W + 100, H + 70], [-200, -200, W + 200, H + 200], [100, 100, 100, 100], [120, 90, 110, 80],\n"
        "           [0, 0, W - 16, H - 16], [16, 32, 16 + 28 * 16, 32 + 28 * 16], [50.3, 60.7, 50.9, 61.2], [0, 0, 448, 448], [W - 1, H - 1, W, H],\n"
        "           [-1000, 10, -900, 60], [10, H + 500, 90, H + 600]]\n"
        "rois = np.zeros((B, R, 4), np.float32)\n"
        "for b in range(B):\n"
        "    for r in range(R):\n"
        "        if r < len(special): rois[b, r] = special[(r + b) % len(special)]\n"
        "        else:\n"
        "            wh = rng.uniform(4, 400, 2); xy = rng.uniform(-50, [W, H]); rois[b, r] = [xy[0], xy[1], xy[0] + wh[0], xy[1] + wh[1]]\n"
        "rt = torch.from_numpy(rois).cuda()\n"
        "cnt = torch.tensor([R, 7, 0], dtype=torch.int32, device='cuda')\n"
        "out = ops.NHWC(torch.full((B * R, 14, 14, C), -3.0, device='cuda'), c=C)\n"
        "_lib.check(lib.premvos_roi_align_f32(fm.ptr, fm.ps, B, fh, fw, C, rt.data_ptr(), cnt.data_ptr(), R, 1.0 / 16, 14, out.ptr, out.ps, _lib.current_stream()), 'roi')\n"
        "torch.cuda.synchronize()\n"
        "np.save(sys.argv[1], out.buf.cpu().numpy())\n")
    outs = []
    for mode in ("bin", "row"):
        f = tmp_path / f"{mode}.npy"
        r = subprocess.run([sys.executable, str(script), str(f)], env=dict(os.environ, PREMVOS_ROI_ALIGN=mode), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(f))
    assert np.isfinite(outs[0]).all() and not (outs[0] == -3.0).any()                 # every slot written (invalid RoIs: zeros)
    assert np.array_equal(outs[0].view(np.int32), outs[1].view(np.int32))             # bit patterns
    assert float(np.abs(outs[0][24 + 7:48]).max()) == 0.0 and float(np.abs(outs[0][48:]).max()) == 0.0    # beyond an image's count: zeros
