"""Generates tests/golden/proposal_ref_*.npz / proposal_host_refs.json by IMPORTING AND EXECUTING the reference's proposal_net
python in the build container (where /root/reference exists): config.py, data.py, common.py, eval.py, basemodel.py,
model.py and train.py's Model._build_graph / convert_results_to_json run unmodified; TensorFlow 1.8, tensorpack, cv2 and
pycocotools -- absent from the image -- are replaced by tools/tfshim.py (an eager numpy/torch stand-in whose primitives are
a restatement of the published TF semantics; see its header for exactly what that does and does not pin).

Fixtures are data only (inputs are regenerated from premvos_amd.synth seeds by the tests):
  proposal_host_refs.json    config constants, CustomResize shapes, clip_boxes, detect_one_image + convert_results_to_json
                             on a fake predictor, the list of variables (name, shape) the graph code requests
  proposal_ref_anchors.npz   data.get_all_anchors(): the full 83x83x15x4 field
  proposal_ref_graph.npz     one inference pass of Model._build_graph on a 112x160 image, ResNet depth (1,1,2,1):
                             featuremap, RPN logits / deltas, proposals, RoIAlign output / conv5 feature (sub-sampled), head logits, final detections
  proposal_ref_mask.npz      the same pass with config.MODE_MASK = True: final_masks [M,14,14] (+ the boxes / labels they belong to)
  proposal_ref_boxops.npz    decode_bbox_target / clip_boxes / generate_rpn_proposals / roi_align / fastrcnn_predictions on
                             seeded tensors incl. score ties

Usage: python tools/make_golden_tf.py [/root/reference]"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True

import tfshim  # noqa: E402

tf = tfshim.install(is_training=False)
sys.path.insert(0, os.path.join(REF, "code", "proposal_net"))
for stub in ("viz", "forward_proto", "hypotheses_pb2"):          # reference modules off the hot path (drawing, protobuf IO)
    sys.modules[stub] = tfshim._Anything(stub)

os.environ.setdefault("USER", "nobody")          # config.py:19 builds a dataset path from it
import config  # noqa: E402   (the reference's)

# what `train.py --forward ... --agnostic --second_head` sets before building the graph (train.py:591-637)
config.CATEGORY_AGNOSTIC = True
import coco  # noqa: E402,F401  (sets NUM_CLASS / SECOND_NUM_CLASS at import, coco.py:20-25)
config.NUM_CLASS = 2
config.USE_SECOND_HEAD = True
config.MODE_MASK = False
import common  # noqa: E402
import data  # noqa: E402
import eval as ref_eval  # noqa: E402
import model as ref_model  # noqa: E402
import train  # noqa: E402

T = tfshim.T
BLOCKS = (1, 1, 2, 1)
IMG_H, IMG_W = 112, 160


def host_refs():
    g = {}
    g["config"] = {k: (v.tolist() if isinstance(v, np.ndarray) else float(v) if isinstance(v, (np.floating, float)) else v)
                   for k, v in vars(config).items()
                   if k.isupper() and isinstance(v, (int, float, tuple, list, np.ndarray, np.floating, bool))}
    shapes = [(480, 854), (1080, 1920), (854, 480), (600, 600), (375, 1242), (100, 1000), (801, 1334), (1333, 800),
              (13, 17), (720, 1280), (1200, 1200), (480, 853), (481, 854)]
    rs = common.CustomResize(config.SHORT_EDGE_SIZE, config.MAX_SIZE)
    g["custom_resize"] = []
    for h, w in shapes:
        t = rs._get_augment_params(np.zeros((h, w, 3), np.uint8))
        g["custom_resize"].append({"h": h, "w": w, "newh": int(t.newh), "neww": int(t.neww)})
    rng = np.random.default_rng(11)
    boxes = rng.uniform(-60, 900, (12, 4)).astype(np.float32)
    g["clip_boxes"] = {"boxes": boxes.tolist(), "shape": [480, 854],
                       "out": common.clip_boxes(boxes.copy(), (480, 854)).tolist()}
    # eval.detect_one_image on a canned predictor + train.convert_results_to_json: scale, un-scale, clip, rounding
    dets = []
    for (h, w), n in (((480, 854), 5), ((1080, 1920), 3), ((333, 500), 0)):
        r = np.random.default_rng(h)
        nh, nw = (lambda t: (t.newh, t.neww))(rs._get_augment_params(np.zeros((h, w, 3), np.uint8)))
        xy = r.uniform(-20, [nw * 0.8, nh * 0.8], (n, 2))
        fb = np.concatenate([xy, xy + r.uniform(5, [nw * 0.5, nh * 0.5], (n, 2))], 1).astype(np.float32)
        fp = r.uniform(0.5, 1, n).astype(np.float32)
        fake = (fb, fp, np.ones(n, np.int64), np.stack([1 - fp, fp], 1), np.full(n, 3, np.int64),
                r.random((n, 81)).astype(np.float32))
        res = ref_eval.detect_one_image(np.zeros((h, w, 3), np.uint8), lambda img, fake=fake: tuple(a.copy() for a in fake))
        boxes_out = [np.array(x.box).tolist() for x in res]
        js = train.convert_results_to_json(res, 0)
        dets.append({"h": h, "w": w, "resized": [int(nh), int(nw)], "final_boxes": fb.tolist(), "final_probs": fp.tolist(),
                     "boxes_after_detect": boxes_out, "json": js})
    g["detect_and_json"] = dets
    g["output_names"] = train.get_model_output_names()
    return g


def tf_variables(weights):
    """premvos_amd.synth's name -> tensor dict laid out as the TF checkpoint would be (premvos_amd.weights writer maps)."""
    from premvos_amd import weights as W
    return {k: np.asarray(v) for k, v in W.proposal_weights_to_tf(weights).items()}


def run_graph():
    from premvos_amd import synth
    w = synth.proposal_weights(3, BLOCKS)
    tfshim.VARIABLES.clear()
    tfshim.VARIABLES.update(tf_variables(w))
    tfshim.REQUESTED.clear()
    tfshim.NAMED.clear()
    config.RESNET_NUM_BLOCK = list(BLOCKS)
    fr, _ = synth.video_frames(1, IMG_H, IMG_W, rank=7)
    img = fr[0].numpy()[:, :, ::-1].astype(np.float32)            # the predictor is fed the resized BGR image as float32
    cap = {}

    def capture(name, fn):
        def wrapped(*a, **k):
            out = fn(*a, **k)
            cap[name] = out
            return out
        return wrapped
    for name in ("pretrained_resnet_conv4", "rpn_head", "generate_rpn_proposals", "roi_align", "resnet_conv5",
                 "fastrcnn_head", "secondclassification_head"):
        setattr(train, name, capture(name, getattr(train, name)))
    fh, fw = IMG_H // config.ANCHOR_STRIDE, IMG_W // config.ANCHOR_STRIDE
    inputs = [T(img), T(np.zeros((fh, fw, config.NUM_ANCHOR), np.int32)), T(np.ones((fh, fw, config.NUM_ANCHOR, 4), np.float32)),
              T(np.zeros((0, 4), np.float32)), T(np.zeros((0,), np.int64)), T(np.zeros((0,), np.int64))]
    if config.MODE_MASK:                                   # train.py:111: one more (unused at inference) input, gt_masks
        inputs.append(T(np.zeros((0, IMG_H, IMG_W), np.uint8)))
    train.Model()._build_graph(inputs)
    a = lambda t: np.asarray(t.a)                                                     # noqa: E731
    out = {"image_bgr_f32": img, "featuremap": a(cap["pretrained_resnet_conv4"]),
           "rpn_label_logits": a(cap["rpn_head"][0]), "rpn_box_logits": a(cap["rpn_head"][1]),
           "proposal_boxes": a(cap["generate_rpn_proposals"][0]), "proposal_scores": a(cap["generate_rpn_proposals"][1]),
           # the two big tensors are stored sub-sampled (every 5th RoI, every 64th channel) + the pooled conv5 feature
           "roi_resized_sub": a(cap["roi_align"])[::5, ::64], "feature_fastrcnn_sub": a(cap["resnet_conv5"])[::5, ::64],
           "feature_fastrcnn_pooled": a(cap["resnet_conv5"]).mean(axis=(2, 3), dtype=np.float32)[:, ::4],
           "fastrcnn_label_logits": a(cap["fastrcnn_head"][0]), "fastrcnn_box_logits": a(cap["fastrcnn_head"][1]),
           "second_label_logits": a(cap["secondclassification_head"])}
    for k in train.get_model_output_names():
        out[k] = a(tfshim.NAMED[k])
    out["fastrcnn_all_boxes"] = a(tfshim.NAMED["fastrcnn_all_boxes"])
    return out, list(tfshim.REQUESTED)


def run_mask_graph():
    """The same pass with MODE_MASK on (train.py:297-309, model.py:494-509): RoIAlign on the final boxes -> conv5 (the shared
    weights) -> Deconv2D 2x2 s2 + ReLU -> 1x1 -> per-label gather -> sigmoid.  Off in --forward (train.py:636-637)."""
    config.MODE_MASK = True
    try:
        out, requested = run_graph()
    finally:
        config.MODE_MASK = False
    return {"final_masks": np.asarray(tfshim.NAMED["final_masks"].a), "final_boxes": out["final_boxes"],
            "final_labels": out["final_labels"], "final_probs": out["final_probs"]}, \
        [r for r in requested if r[0].startswith("maskrcnn/")]


def box_ops():
    """The pure box arithmetic of model.py on seeded tensors, incl. exact score ties and degenerate boxes."""
    r = np.random.default_rng(21)
    g = {}
    anchors = data.get_all_anchors()[:6, :9].reshape(-1, 4)
    deltas = (r.standard_normal(anchors.shape) * [0.3, 0.3, 1.5, 1.5]).astype(np.float32)
    deltas[5, 2:] = 9.0                                         # beyond BBOX_DECODE_CLIP
    dec = ref_model.decode_bbox_target(T(deltas), T(anchors))
    g["decode_anchors"], g["decode_deltas"], g["decode_out"] = anchors, deltas, dec.a
    g["clip_out"] = ref_model.clip_boxes(dec, T(np.array([70, 120], np.int32))).a
    scores = r.standard_normal(len(anchors)).astype(np.float32)
    scores[10:20] = scores[10]                                  # ties
    scores[100:104] = scores[3]
    pb, ps = ref_model.generate_rpn_proposals(dec, T(scores), T(np.array([70, 120], np.int32)))
    g["rpn_scores"], g["rpn_boxes_out"], g["rpn_scores_out"] = scores, pb.a, ps.a
    fm = r.standard_normal((1, 6, 9, 13)).astype(np.float32)
    rois = np.array([[0.2, 0.3, 7.9, 5.1], [-1.5, -2.0, 4.0, 3.0], [10.0, 6.0, 14.5, 9.7], [3.0, 3.0, 3.0, 3.0],
                     [0, 0, 12, 8]], np.float32)
    g["roi_fm"], g["roi_boxes"], g["roi_out"] = fm, rois, ref_model.roi_align(T(fm), T(rois), 7).a
    n = 40
    xy = r.uniform(0, 80, (n, 2))
    boxes = np.concatenate([xy, xy + r.uniform(4, 60, (n, 2))], 1).astype(np.float32).reshape(n, 1, 4)
    boxes[7] = boxes[3]
    p1 = r.uniform(0.3, 1.0, n).astype(np.float32)
    p1[7] = p1[3]
    p1[20:23] = 0.75
    probs = np.stack([1 - p1, p1], 1)
    sel, tp = ref_model.fastrcnn_predictions(T(boxes), T(probs))
    g["pred_boxes"], g["pred_probs"], g["pred_selection"], g["pred_topk_probs"] = boxes, probs, sel.a, tp.a
    return g


def main():
    os.makedirs(GOLD, exist_ok=True)
    hr = host_refs()
    np.savez_compressed(os.path.join(GOLD, "proposal_ref_anchors.npz"), anchors=data.get_all_anchors())
    graph, requested = run_graph()
    hr["graph"] = {"blocks": list(BLOCKS), "image_hw": [IMG_H, IMG_W], "weights": "premvos_amd.synth.proposal_weights(3, blocks)",
                   "image": "premvos_amd.synth.video_frames(1, h, w, rank=7)[0][0] as BGR float32",
                   "variables": [[n, list(s)] for n, s in requested]}
    np.savez_compressed(os.path.join(GOLD, "proposal_ref_graph.npz"), **graph)
    np.savez_compressed(os.path.join(GOLD, "proposal_ref_boxops.npz"), **box_ops())
    mask, mask_vars = run_mask_graph()
    hr["graph"]["mask_variables"] = [[n, list(s)] for n, s in mask_vars]
    np.savez_compressed(os.path.join(GOLD, "proposal_ref_mask.npz"), **mask)
    with open(os.path.join(GOLD, "proposal_host_refs.json"), "w") as f:
        json.dump(hr, f, indent=1)
    for fn in ("proposal_ref_anchors.npz", "proposal_ref_graph.npz", "proposal_ref_boxops.npz", "proposal_ref_mask.npz",
               "proposal_host_refs.json"):
        print(fn, os.path.getsize(os.path.join(GOLD, fn)), "bytes")
    print("final detections:", len(graph["final_probs"]), "proposals:", len(graph["proposal_scores"]))


if __name__ == "__main__":
    main()
