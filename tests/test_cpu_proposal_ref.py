"""oracle/proposal_oracle.py and the product's host twins against fixtures produced by EXECUTING the reference's own
proposal_net python (tools/make_golden_tf.py: config.py, data.py, common.py, eval.py, basemodel.py, model.py and
train.py's Model._build_graph / convert_results_to_json, unmodified, on tools/tfshim.py's eager stand-in for TF 1.8 /
tensorpack).  What that pins: constants, anchors, resize shape math, box arithmetic, the composition of the whole
inference graph, the variable names the graph asks its checkpoint for.  The TF primitives themselves (conv, NMS,
crop_and_resize ...) are restated in the stand-in and remain third-party semantics."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import proposal_oracle as PO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HR = json.load(open(os.path.join(GOLD, "proposal_host_refs.json")))
GRAPH = np.load(os.path.join(GOLD, "proposal_ref_graph.npz"))
BOX = np.load(os.path.join(GOLD, "proposal_ref_boxops.npz"))
ANCH = np.load(os.path.join(GOLD, "proposal_ref_anchors.npz"))["anchors"]


def test_config_constants_of_oracle_and_product():
    from premvos_amd.proposal import driver as D
    from premvos_amd.proposal import model as M
    c = HR["config"]
    for mod in (PO, M):
        assert tuple(mod.RESNET_NUM_BLOCK) == tuple(c["RESNET_NUM_BLOCK"])
        assert mod.ANCHOR_STRIDE == c["ANCHOR_STRIDE"] and tuple(mod.ANCHOR_SIZES) == tuple(c["ANCHOR_SIZES"])
        assert tuple(mod.ANCHOR_RATIOS) == tuple(c["ANCHOR_RATIOS"]) and mod.NUM_ANCHOR == c["NUM_ANCHOR"]
        assert np.float32(mod.BBOX_DECODE_CLIP) == np.float32(c["BBOX_DECODE_CLIP"])
        assert mod.RPN_PROPOSAL_NMS_THRESH == c["RPN_PROPOSAL_NMS_THRESH"] and mod.RPN_MIN_SIZE == c["RPN_MIN_SIZE"]
        assert mod.TEST_PRE_NMS_TOPK == c["TEST_PRE_NMS_TOPK"] and mod.TEST_POST_NMS_TOPK == c["TEST_POST_NMS_TOPK"]
        assert mod.FASTRCNN_NMS_THRESH == c["FASTRCNN_NMS_THRESH"] and mod.RESULT_SCORE_THRESH == c["RESULT_SCORE_THRESH"]
        assert mod.RESULTS_PER_IM == c["RESULTS_PER_IM"] and mod.NUM_CLASS == c["NUM_CLASS"] == 2
        assert mod.SECOND_NUM_CLASS == c["SECOND_NUM_CLASS"] == 81
        assert [float(v) for v in mod.FASTRCNN_BBOX_REG_WEIGHTS] == c["FASTRCNN_BBOX_REG_WEIGHTS"]
    assert (PO.SHORT_EDGE_SIZE, PO.MAX_SIZE) == (D.SHORT_EDGE_SIZE, D.MAX_SIZE) == (c["SHORT_EDGE_SIZE"], c["MAX_SIZE"])
    assert HR["output_names"] == ["final_boxes", "final_probs", "final_labels", "final_posterior", "second_final_labels",
                                  "second_final_posterior"]


def test_custom_resize_shapes():
    from premvos_amd.proposal.driver import custom_resize_shape
    for r in HR["custom_resize"]:
        assert PO.custom_resize_shape(r["h"], r["w"]) == (r["newh"], r["neww"]), r
        assert custom_resize_shape(r["h"], r["w"]) == (r["newh"], r["neww"]), r


def test_anchor_field_full():
    """data.get_all_anchors() (83 x 83 x 15 x 4, incl. the +1 on x2/y2) against the oracle and the product's cell table."""
    from premvos_amd.proposal import cell_anchors
    assert ANCH.shape == (83, 83, 15, 4) and ANCH.dtype == np.float32
    for fh, fw in ((83, 83), (47, 83), (46, 83), (7, 10)):
        assert np.array_equal(PO.all_anchors(fh, fw), ANCH[:fh, :fw])
    assert np.array_equal(cell_anchors(), ANCH[0, 0]) and np.array_equal(PO.cell_anchors(), ANCH[0, 0])
    # the kernel forms anchor (y, x, a) as cell + 16*(x, y, x, y) in float32: exact for this field
    ys, xs = np.meshgrid(np.arange(83, dtype=np.float32) * 16, np.arange(83, dtype=np.float32) * 16, indexing="ij")
    shift = np.stack([xs, ys, xs, ys], -1)[:, :, None, :]
    assert np.array_equal(cell_anchors()[None, None] + shift, ANCH)


def test_clip_boxes_and_detect_one_image_json():
    from premvos_amd.proposal.driver import _to_results, clip_boxes, convert_results_to_json
    cb = HR["clip_boxes"]
    got = clip_boxes(np.array(cb["boxes"], np.float32), tuple(cb["shape"]))
    assert np.array_equal(got, np.array(cb["out"], np.float32))
    for d in HR["detect_and_json"]:
        n = len(d["final_probs"])
        fb = np.array(d["final_boxes"], np.float32).reshape(n, 4)
        fp = np.array(d["final_probs"], np.float32)
        nh, nw = d["resized"]
        scale = (nh * 1.0 / d["h"] + nw * 1.0 / d["w"]) / 2                     # eval.py:78
        res = _to_results(fb.copy(), fp, np.ones(n, np.int64), np.stack([1 - fp, fp], 1), np.full(n, 3, np.int64),
                          np.zeros((n, 81), np.float32), scale, (d["h"], d["w"]))
        assert [np.array(r.box).tolist() for r in res] == d["boxes_after_detect"]
        assert convert_results_to_json(res) == d["json"]                       # incl. the float32 digits of round(x, 1)
        # oracle twin
        ob = fb / scale
        ob = PO_clip(ob, d["h"], d["w"])
        assert PO.results_to_json(ob, fp) == d["json"]


def PO_clip(b, h, w):
    b = b.copy()
    b[:, [0, 1]] = np.maximum(b[:, [0, 1]], 0)
    b[:, 2] = np.minimum(b[:, 2], w)
    b[:, 3] = np.minimum(b[:, 3], h)
    return b


def test_box_arithmetic_of_model_py():
    """decode_bbox_target (model.py:113-139), clip_boxes (:17-27), generate_rpn_proposals (:169-217) on seeded tensors
    with exact score ties and a delta beyond BBOX_DECODE_CLIP."""
    dec = PO.decode_bbox_target(BOX["decode_deltas"], BOX["decode_anchors"])
    assert np.abs(dec - BOX["decode_out"]).max() <= 1e-4 * np.abs(BOX["decode_out"]).max()      # np.exp vs torch/TF exp ulps
    assert np.array_equal(PO.clip_boxes(BOX["decode_out"], 70, 120), BOX["clip_out"])
    pb, ps, pidx = PO.generate_rpn_proposals(BOX["decode_out"], BOX["rpn_scores"], 70, 120)
    assert np.array_equal(ps, BOX["rpn_scores_out"]) and np.array_equal(pb, BOX["rpn_boxes_out"])
    assert np.array_equal(BOX["rpn_scores"][pidx], ps)


def test_roi_align_of_model_py():
    got = PO.roi_align(torch.from_numpy(BOX["roi_fm"]), BOX["roi_boxes"], 7).numpy()
    assert got.shape == BOX["roi_out"].shape
    assert np.abs(got - BOX["roi_out"]).max() < 1e-5


def test_fastrcnn_predictions_of_model_py():
    """fastrcnn_predictions (model.py:438-491): threshold, per-class NMS, top RESULTS_PER_IM -- indices in the golden's order."""
    boxes, probs = BOX["pred_boxes"], BOX["pred_probs"]
    sel, tp = BOX["pred_selection"], BOX["pred_topk_probs"]                    # sel: [k, 2] = (box id, category id)
    # the oracle's tail takes logits + deltas; feed it logits whose softmax is `probs` and zero deltas on the given boxes
    logits = np.log(probs.astype(np.float64)).astype(np.float32)
    h = w = 10000                                                                # no clipping
    fb, fp, fl, fi = PO.fastrcnn_tail(logits, np.zeros((len(boxes), 1, 4), np.float32), boxes[:, 0], h, w)
    assert np.array_equal(fi, sel[:, 0])
    assert np.abs(fp - tp).max() < 1e-6


def _synth_inputs():
    from premvos_amd import synth
    g = HR["graph"]
    w = synth.proposal_weights(3, tuple(g["blocks"]))
    fr, _ = synth.video_frames(1, g["image_hw"][0], g["image_hw"][1], rank=7)
    img = np.ascontiguousarray(fr[0].numpy()[:, :, ::-1])
    assert np.array_equal(img.astype(np.float32), GRAPH["image_bgr_f32"])
    return w, img, tuple(g["blocks"])


def _close(a, b, tol):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() <= tol * max(1.0, float(np.abs(b).max()))


def test_whole_graph_oracle_vs_reference_build_graph():
    """One inference pass of the reference's Model._build_graph (train.py:107-309) vs oracle.model_forward, stage by stage."""
    w, img, blocks = _synth_inputs()
    (fb, fp, fl, fi), it = PO.model_forward(w, img, blocks, intermediates=True)
    assert _close(it["featuremap"].numpy(), GRAPH["featuremap"], 1e-4)
    assert _close(it["rpn_logits"].numpy(), GRAPH["rpn_label_logits"], 1e-4)
    assert _close(it["rpn_box"].numpy(), GRAPH["rpn_box_logits"], 1e-4)
    assert len(it["proposals"]) == len(GRAPH["proposal_scores"])
    assert _close(it["proposal_scores"], GRAPH["proposal_scores"], 1e-4)
    assert np.abs(it["proposals"] - GRAPH["proposal_boxes"]).max() < 1e-2
    assert _close(it["roi"].numpy()[::5, ::64], GRAPH["roi_resized_sub"], 1e-4)
    assert _close(it["feat5"].numpy()[::5, ::64], GRAPH["feature_fastrcnn_sub"], 1e-4)
    assert _close(it["feat5"].numpy().mean(axis=(2, 3))[:, ::4], GRAPH["feature_fastrcnn_pooled"], 1e-4)
    assert _close(it["cls"].numpy(), GRAPH["fastrcnn_label_logits"], 1e-4)
    assert _close(it["box"].numpy(), GRAPH["fastrcnn_box_logits"], 1e-4)
    assert _close(it["second"].numpy(), GRAPH["second_label_logits"], 1e-4)
    assert len(fp) == len(GRAPH["final_probs"])
    assert np.abs(fp - GRAPH["final_probs"]).max() < 1e-5 and np.abs(fb - GRAPH["final_boxes"]).max() < 1e-2
    assert np.array_equal(fl, GRAPH["final_labels"])


def test_mask_head_oracle_vs_reference_build_graph():
    """The same pass with config.MODE_MASK on (train.py:297-309, model.py:494-509, tensorpack Deconv2D): final_masks."""
    ref = np.load(os.path.join(GOLD, "proposal_ref_mask.npz"))
    assert np.array_equal(ref["final_boxes"], GRAPH["final_boxes"])              # the mask branch changes nothing upstream
    w, img, blocks = _synth_inputs()
    (fb, fp, fl, fi), it = PO.model_forward(w, img, blocks, intermediates=True)
    masks = PO.maskrcnn_masks(w, it["featuremap"], ref["final_boxes"], blocks[3])
    assert masks.shape == ref["final_masks"].shape == (len(fp), 14, 14)
    assert np.abs(masks - ref["final_masks"]).max() < 1e-4
    # ... and the checkpoint names / TF layouts the branch asks for are the importer's (Deconv2D W = [kh, kw, out, in])
    from premvos_amd import synth
    from premvos_amd import weights as W
    tfv = W.proposal_weights_to_tf(synth.proposal_weights(3, blocks))
    for n, shp in HR["graph"]["mask_variables"]:
        assert tuple(tfv[n].shape) == tuple(shp), n


def test_variable_names_the_graph_requests_are_the_importers():
    """Every variable Model._build_graph asked its checkpoint for (name + TF-layout shape) is what
    premvos_amd.weights.proposal_weights_from_tf consumes, and the mapped dict has every key ProposalNet packs."""
    from premvos_amd import synth
    from premvos_amd import weights as W
    req = {n: tuple(s) for n, s in HR["graph"]["variables"]}
    w = synth.proposal_weights(3, tuple(HR["graph"]["blocks"]))
    tfv = W.proposal_weights_to_tf(w)
    missing = [n for n in req if n not in tfv]
    assert not missing, missing
    assert all(tuple(tfv[n].shape) == s for n, s in req.items())
    unused = sorted(set(tfv) - set(req))
    assert all(n.startswith("maskrcnn/") for n in unused), unused        # the mask head is off in --forward
    back = W.proposal_weights_from_tf({n: tfv[n] for n in req})
    need = [k for k in w if not k.startswith("maskrcnn/")]
    assert sorted(back) == sorted(need)
    for k in need:
        a, b = w[k], back[k]
        if isinstance(a, dict):
            assert all(torch.equal(a[j], b[j]) for j in a)
        else:
            assert torch.equal(a, b), k
