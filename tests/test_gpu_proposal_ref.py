"""The HIP proposal path against fixtures produced by EXECUTING the reference's proposal_net python (tools/make_golden_tf.py,
see tests/test_cpu_proposal_ref.py for what that pins): the whole inference graph of train.py:107-309 on a 112x160 image, and
model.py's box arithmetic on seeded tensors with score ties."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HR = json.load(open(os.path.join(GOLD, "proposal_host_refs.json")))
GRAPH = np.load(os.path.join(GOLD, "proposal_ref_graph.npz"))
BOX = np.load(os.path.join(GOLD, "proposal_ref_boxops.npz"))


def _close(a, b, tol):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() <= tol * max(1.0, float(np.abs(b).max()))


def test_whole_graph_hip_vs_reference_build_graph():
    from premvos_amd import synth
    from premvos_amd.proposal import OfflinePredictor, ProposalNet
    g = HR["graph"]
    blocks = tuple(g["blocks"])
    w = synth.proposal_weights(3, blocks)
    fr, _ = synth.video_frames(1, g["image_hw"][0], g["image_hw"][1], rank=7)
    img = np.ascontiguousarray(fr[0].numpy()[:, :, ::-1])
    net = ProposalNet(w, blocks, use_graph=False)
    out = OfflinePredictor(net)(img)
    p = net.plan(1, *g["image_hw"])
    assert _close(p.featuremap.torch().cpu().numpy(), GRAPH["featuremap"], 1e-3)
    rpn = p.rpn_out.buf[0].cpu().numpy()
    assert _close(rpn[:, :, :15], GRAPH["rpn_label_logits"], 1e-3)
    assert _close(rpn[:, :, 15:75].reshape(GRAPH["rpn_box_logits"].shape), GRAPH["rpn_box_logits"], 1e-3)
    n = int(p.roi_count.item())
    assert n == len(GRAPH["proposal_scores"])
    # same anchors in the same order: the reference's scores are a permutation-free match at 1e-3, boxes within 0.05 px
    assert np.abs(p.roi_scores[0, :n].cpu().numpy() - GRAPH["proposal_scores"]).max() < 5e-3
    assert np.abs(p.rois[0, :n].cpu().numpy() - GRAPH["proposal_boxes"]).max() < 0.05
    f5 = p.feat5.torch().cpu().numpy()[:n]
    assert _close(f5[::5, ::64], GRAPH["feature_fastrcnn_sub"], 2e-3)
    boxes, probs, labels, post, sl, sp = out[:6]
    assert len(probs) == len(GRAPH["final_probs"])
    assert np.abs(probs - GRAPH["final_probs"]).max() < 1e-4 and np.abs(boxes - GRAPH["final_boxes"]).max() < 0.05
    assert np.array_equal(labels, GRAPH["final_labels"]) and np.array_equal(sl, GRAPH["second_final_labels"])
    assert np.abs(post - GRAPH["final_posterior"]).max() < 1e-4
    assert np.abs(sp - GRAPH["second_final_posterior"]).max() < 1e-3


def test_mask_head_hip_vs_reference_build_graph():
    """MODE_MASK branch (train.py:297-309, model.py:494-509) of the reference-executed graph: final_masks [M,14,14]."""
    from premvos_amd import synth
    from premvos_amd.proposal import OfflinePredictor, ProposalNet
    ref = np.load(os.path.join(GOLD, "proposal_ref_mask.npz"))
    g = HR["graph"]
    blocks = tuple(g["blocks"])
    w = synth.proposal_weights(3, blocks)
    fr, _ = synth.video_frames(1, g["image_hw"][0], g["image_hw"][1], rank=7)
    img = np.ascontiguousarray(fr[0].numpy()[:, :, ::-1])
    net = ProposalNet(w, blocks, use_graph=False, mode_mask=True)
    out = OfflinePredictor(net)(img)
    assert np.abs(out[0] - ref["final_boxes"]).max() < 0.05 and len(out[1]) == len(ref["final_probs"])
    masks = net.masks(net.plan(1, *g["image_hw"]))
    assert masks.shape == ref["final_masks"].shape
    assert np.abs(masks - ref["final_masks"]).max() < 2e-3


def test_rpn_proposals_kernel_on_reference_box_arithmetic():
    """generate_rpn_proposals(decode_bbox_target(...)) of model.py:113-217 on seeded deltas with score ties: the fused kernel
    must return the reference's boxes and scores in the reference's order."""
    from premvos_amd import _lib, ops
    from premvos_amd.proposal import cell_anchors
    fh, fw = 6, 9
    deltas, scores = BOX["decode_deltas"], BOX["rpn_scores"]
    rpn = ops.NHWC.alloc(1, fh, fw, 75)
    rpn.buf[0, :, :, :15] = torch.from_numpy(scores.reshape(fh, fw, 15)).cuda()
    rpn.buf[0, :, :, 15:75] = torch.from_numpy(deltas.reshape(fh, fw, 60)).cuda()
    ca = torch.from_numpy(cell_anchors()).cuda()
    ob = torch.zeros((1, 100, 4), device="cuda")
    osc = torch.zeros((1, 100), device="cuda")
    oi = torch.zeros((1, 100), dtype=torch.int32, device="cuda")
    oc = torch.zeros((1,), dtype=torch.int32, device="cuda")
    _lib.check(_lib.load().premvos_rpn_proposals_f32(
        rpn.ptr, rpn.ps, 1, fh, fw, 15, 0, 15, ca.data_ptr(), 16.0, 70.0, 120.0, 1000, 100, 0.7, 0.0,
        float(HR["config"]["BBOX_DECODE_CLIP"]), ob.data_ptr(), osc.data_ptr(), oi.data_ptr(), oc.data_ptr(),
        _lib.current_stream()))
    n = int(oc.item())
    assert n == len(BOX["rpn_scores_out"])
    assert np.array_equal(osc[0, :n].cpu().numpy(), BOX["rpn_scores_out"])
    assert np.array_equal(scores[oi[0, :n].cpu().numpy()], BOX["rpn_scores_out"])
    assert np.abs(ob[0, :n].cpu().numpy() - BOX["rpn_boxes_out"]).max() < 1e-3


def test_roi_align_kernel_on_reference_roi_align():
    from premvos_amd import _lib, ops
    fm, rois, ref = BOX["roi_fm"], BOX["roi_boxes"], BOX["roi_out"]
    _, c, H, W = fm.shape
    cp = (c + 3) // 4 * 4                     # the kernel moves 4 channels per lane: pad the 6-channel fixture with zeros
    f = ops.NHWC.alloc(1, H, W, cp)
    f.buf[0, :, :, :c] = torch.from_numpy(fm[0]).permute(1, 2, 0).cuda()
    n = len(rois)
    r = torch.zeros((1, 8, 4), device="cuda")
    r[0, :n] = torch.from_numpy(rois).cuda()
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    out = ops.NHWC.alloc(8, 7, 7, cp)
    # model.py's roi_align takes boxes already in feature-map coordinates: spatial_scale 1
    _lib.check(_lib.load().premvos_roi_align_f32(f.ptr, f.ps, 1, H, W, cp, r.data_ptr(), cnt.data_ptr(), 8, 1.0, 7, out.ptr,
                                                 out.ps, _lib.current_stream()))
    got = out.torch().cpu().numpy()[:n, :c]
    assert np.abs(got - ref).max() < 1e-5
    assert np.abs(out.torch().cpu().numpy()[:n, c:]).max() == 0


def test_frcnn_tail_kernel_on_reference_fastrcnn_predictions():
    from premvos_amd import _lib
    boxes, probs = BOX["pred_boxes"][:, 0], BOX["pred_probs"]
    sel, tp = BOX["pred_selection"], BOX["pred_topk_probs"]
    n = len(boxes)
    head = torch.zeros((n, 8), device="cuda")
    head[:, :2] = torch.from_numpy(np.log(probs.astype(np.float64)).astype(np.float32)).cuda()      # zero deltas
    rois = torch.from_numpy(boxes).cuda().view(1, n, 4).contiguous()
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    ob = torch.zeros((1, 20, 4), device="cuda")
    op = torch.zeros((1, 20), device="cuda")
    oi = torch.zeros((1, 20), dtype=torch.int32, device="cuda")
    oc = torch.zeros((1,), dtype=torch.int32, device="cuda")
    _lib.check(_lib.load().premvos_frcnn_tail_f32(head.data_ptr(), 8, rois.data_ptr(), cnt.data_ptr(), 1, n, 10000.0, 10000.0,
                                                  0.5, 0.5, 20, float(HR["config"]["BBOX_DECODE_CLIP"]), 10.0, 10.0, 5.0, 5.0,
                                                  ob.data_ptr(), op.data_ptr(), oi.data_ptr(), oc.data_ptr(),
                                                  _lib.current_stream()))
    m = int(oc.item())
    assert m == len(tp)
    assert np.array_equal(oi[0, :m].cpu().numpy(), sel[:, 0].astype(np.int32))
    assert np.abs(op[0, :m].cpu().numpy() - tp).max() < 1e-6
    assert np.abs(ob[0, :m].cpu().numpy() - boxes[sel[:, 0]]).max() < 1e-3
