"""CPU oracle for the proposal_net forward path (class-agnostic ResNet-101-C4 Faster R-CNN, --forward).
TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Plain PyTorch/numpy fp32 restatement of the reference graph (paths relative to code/proposal_net/):

* image preprocess .................... basemodel.py:12-26, train.py:87-90
* ResNet-101 C4 backbone .............. basemodel.py:29-89 (explicit asymmetric pads :55-56,79-82, shortcut crop :41)
* RPN head ............................ model.py:30-51
* anchors ............................. data.py:34-74, utils/generate_anchors.py:40-100, train.py:92-105
* box decode / clip ................... model.py:113-139, 17-27
* proposal top-k / filter / NMS ....... model.py:169-217
* RoIAlign ............................ model.py:300-374
* conv5 + heads ....................... basemodel.py:92-99, model.py:377-395, 551-565, train.py:164-191
* inference tail ...................... train.py:275-295, model.py:438-491
* host side ........................... eval.py:61-110, common.py:35-62,107-119, train.py:388-428

The arithmetic of Conv2D/BatchNorm/crop_and_resize/non_max_suppression/top_k lives in third-party code that
is NOT under /root/reference: TensorFlow 1.8 (README.md:20) and tensorpack @6fdde15 (proposal_net/README:7-9);
their published semantics are restated here (BatchNorm eps 1e-5, BNReLU = BN then ReLU; crop_and_resize:
in = y1*(H-1) + i*(y2-y1)*(H-1)/(crop-1), extrapolation 0 outside [0,H-1]; NMS: descending score, suppress
when IoU > thresh, boxes as given, area from min/max corners).  Where TF leaves an order unspecified
(top_k(sorted=False), equal scores in NMS) this oracle and the HIP path both use: descending score, ties to
the lower index.

PARITY: the reference holds no test or golden vector for this path (its only in-repo known-answer is the 9-anchor table in
the comments of utils/generate_anchors.py:20-38, reproduced by ``generate_anchors`` below, tests/test_cpu_proposal.py).
PINNED since round 2 by fixtures produced by EXECUTING the reference's own python (config.py, data.py, common.py, eval.py,
basemodel.py, model.py, train.py unmodified; tools/make_golden_tf.py on tools/tfshim.py, an eager stand-in for TF 1.8 /
tensorpack): constants, the full anchor field, resize-shape math, all box arithmetic incl. ties, one whole inference pass of
Model._build_graph (stage by stage, incl. the MODE_MASK branch), JSON rounding, checkpoint variable names
(tests/test_cpu_proposal_ref.py; the HIP twin is tests/test_gpu_proposal_ref.py).  The TF primitives themselves (conv2d,
fused batch norm, top_k, non_max_suppression, crop_and_resize, conv2d_transpose) remain restated: third-party, absent.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# config.py ---------------------------------------------------------------------------------------
RESNET_NUM_BLOCK = (3, 4, 23, 3)        # :61
SHORT_EDGE_SIZE, MAX_SIZE = 800, 1333   # :64-65
ANCHOR_STRIDE = 16                      # :69
ANCHOR_SIZES = (32, 64, 128, 256, 512)  # :71
ANCHOR_RATIOS = (0.5, 1.0, 2.0)         # :72
NUM_ANCHOR = 15
BBOX_DECODE_CLIP = np.float32(np.log(MAX_SIZE / 16.0))   # :77
RPN_MIN_SIZE = 0
RPN_PROPOSAL_NMS_THRESH = 0.7           # :84
TEST_PRE_NMS_TOPK = 1000                # :101
TEST_POST_NMS_TOPK = 100                # :106
FASTRCNN_BBOX_REG_WEIGHTS = np.array([10, 10, 5, 5], np.float32)   # :92
FASTRCNN_NMS_THRESH = 0.5               # :111
RESULT_SCORE_THRESH = 0.5               # :117
RESULTS_PER_IM = 20                     # :123
NUM_CLASS = 2                           # --agnostic (train.py:591-639)
SECOND_NUM_CLASS = 81
BN_EPS = 1e-5                           # tensorpack BatchNorm default


# ---------------------------------------------------------------------------------------------------
# anchors
def _whctrs(a):
    w = a[2] - a[0] + 1
    h = a[3] - a[1] + 1
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def _mk(ws, hs, xc, yc):
    ws, hs = ws[:, None], hs[:, None]
    return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6)) -> np.ndarray:
    """utils/generate_anchors.py:40-100 (ratio enumeration outer, scale inner)."""
    ratios, scales = np.asarray(ratios, np.float64), np.asarray(scales, np.float64)
    base = np.array([1, 1, base_size, base_size], np.float32) - 1
    w, h, xc, yc = _whctrs(base)
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    ratio_anchors = _mk(ws, hs, xc, yc)
    out = []
    for ra in ratio_anchors:
        w, h, xc, yc = _whctrs(ra)
        out.append(_mk(w * scales, h * scales, xc, yc))
    return np.vstack(out)


def cell_anchors() -> np.ndarray:
    """The 15 anchors of feature-map cell (0,0) incl. the ``[:, 2:] += 1`` of data.py:73 -> float32 [15,4]."""
    a = generate_anchors(ANCHOR_STRIDE, ANCHOR_RATIOS, np.array(ANCHOR_SIZES, np.float64) / ANCHOR_STRIDE)
    a = a.astype(np.float32)
    a[:, 2:] += 1
    return a


def all_anchors(fh: int, fw: int) -> np.ndarray:
    """data.py:34-74 sliced to the feature map (train.py:92-105) -> [fh,fw,15,4] float32."""
    ca = cell_anchors()
    sx = (np.arange(fw) * ANCHOR_STRIDE).astype(np.float32)
    sy = (np.arange(fh) * ANCHOR_STRIDE).astype(np.float32)
    shifts = np.stack(np.broadcast_arrays(sx[None, :, None], sy[:, None, None], sx[None, :, None], sy[:, None, None]), -1)
    return (ca[None, None] + shifts.reshape(fh, fw, 1, 4)).astype(np.float32)


# ---------------------------------------------------------------------------------------------------
# synthetic weights (no checkpoint in the build container)
def _g(name: str, seed: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) + 104729 * seed) & 0x7FFFFFFF)


def _conv_w(name, cout, cin, k, seed, gain=2.0):
    return torch.randn((cout, cin, k, k), generator=_g(name, seed)) * math.sqrt(gain / (cin * k * k))


def _bn(name, c, seed, gamma=(0.8, 1.2)):
    g = _g(name, seed)
    return {"gamma": torch.rand(c, generator=g) * (gamma[1] - gamma[0]) + gamma[0],
            "beta": torch.randn(c, generator=g) * 0.1,
            "mean": torch.randn(c, generator=g) * 0.1,
            "var": torch.rand(c, generator=g) + 0.5}


def synth_weights(seed: int = 0, num_blocks=RESNET_NUM_BLOCK) -> Dict[str, object]:
    """Names follow the reference variable scopes: conv0, group{g}/block{i}/conv{1,2,3},
    convshortcut, .../bn/{gamma,beta,mean/EMA,variance/EMA}; rpn/{conv0,class,box}; fastrcnn/{class,box};
    secondclassification/class.  Conv weights are stored OIHW here (TF stores HWIO)."""
    w: Dict[str, object] = {}
    w["conv0/W"] = _conv_w("conv0/W", 64, 3, 7, seed)
    w["conv0/bn"] = _bn("conv0/bn", 64, seed)
    cin = 64
    for g, (feat, cnt) in enumerate(zip((64, 128, 256, 512), num_blocks)):
        for i in range(cnt):
            p = f"group{g}/block{i}"
            w[f"{p}/conv1/W"] = _conv_w(p + "c1", feat, cin, 1, seed)
            w[f"{p}/conv1/bn"] = _bn(p + "b1", feat, seed)
            w[f"{p}/conv2/W"] = _conv_w(p + "c2", feat, feat, 3, seed)
            w[f"{p}/conv2/bn"] = _bn(p + "b2", feat, seed)
            w[f"{p}/conv3/W"] = _conv_w(p + "c3", feat * 4, feat, 1, seed, gain=1.0)
            w[f"{p}/conv3/bn"] = _bn(p + "b3", feat * 4, seed, gamma=(0.15, 0.35))   # keeps the residual sum O(1)
            if cin != feat * 4:
                w[f"{p}/convshortcut/W"] = _conv_w(p + "cs", feat * 4, cin, 1, seed, gain=1.0)
                w[f"{p}/convshortcut/bn"] = _bn(p + "bs", feat * 4, seed)
            cin = feat * 4
    w["rpn/conv0/W"] = _conv_w("rpn/conv0/W", 1024, 1024, 3, seed)
    w["rpn/conv0/b"] = torch.randn(1024, generator=_g("rpn/conv0/b", seed)) * 0.05
    w["rpn/class/W"] = _conv_w("rpn/class/W", NUM_ANCHOR, 1024, 1, seed, gain=8.0)
    w["rpn/class/b"] = torch.randn(NUM_ANCHOR, generator=_g("rpn/class/b", seed)) * 0.5
    w["rpn/box/W"] = _conv_w("rpn/box/W", 4 * NUM_ANCHOR, 1024, 1, seed, gain=0.3)
    w["rpn/box/b"] = torch.randn(4 * NUM_ANCHOR, generator=_g("rpn/box/b", seed)) * 0.1
    cw = torch.randn((NUM_CLASS, 2048), generator=_g("frc", seed)) * 0.4
    w["fastrcnn/class/W"] = cw - cw.mean(dim=1, keepdim=True)      # zero-mean rows: scores spread around 0.5
    w["fastrcnn/class/b"] = torch.randn(NUM_CLASS, generator=_g("frcb", seed)) * 0.1
    w["fastrcnn/box/W"] = torch.randn(((NUM_CLASS - 1) * 4, 2048), generator=_g("frb", seed)) * 0.05
    w["fastrcnn/box/b"] = torch.randn((NUM_CLASS - 1) * 4, generator=_g("frbb", seed)) * 0.1
    # mask head (model.py:494-509): Deconv2D 2x2 s2 -> 256 (+ReLU), Conv2D 1x1 -> NUM_CLASS-1.  [cin,cout,kh,kw] here.
    w["maskrcnn/deconv/W"] = torch.randn((2048, 256, 2, 2), generator=_g("mdc", seed)) * math.sqrt(2.0 / 2048)
    w["maskrcnn/deconv/b"] = torch.randn(256, generator=_g("mdcb", seed)) * 0.05
    w["maskrcnn/conv/W"] = _conv_w("maskrcnn/conv/W", NUM_CLASS - 1, 256, 1, seed, gain=8.0)
    w["maskrcnn/conv/b"] = torch.randn(NUM_CLASS - 1, generator=_g("mcb", seed)) * 0.3
    w["secondclassification/class/W"] = torch.randn((SECOND_NUM_CLASS, 2048), generator=_g("sc", seed)) * 0.05
    w["secondclassification/class/b"] = torch.randn(SECOND_NUM_CLASS, generator=_g("scb", seed)) * 0.1
    return w


# ---------------------------------------------------------------------------------------------------
# backbone
def image_preprocess(img_bgr: torch.Tensor) -> torch.Tensor:
    """[H,W,3] (uint8 or float) BGR -> [1,3,H,W] float32  (basemodel.py:12-26, train.py:87-90)."""
    x = img_bgr.to(torch.float32) * np.float32(1.0 / 255)
    mean = torch.tensor([0.485, 0.456, 0.406][::-1], dtype=torch.float32)
    std = torch.tensor([0.229, 0.224, 0.225][::-1], dtype=torch.float32)
    x = (x - mean) / std
    return x.permute(2, 0, 1).unsqueeze(0).contiguous()


def _bn_apply(x, bn):
    return F.batch_norm(x, bn["mean"], bn["var"], bn["gamma"], bn["beta"], training=False, eps=BN_EPS)


def _bottleneck(w, p, x, feat, stride):
    """basemodel.py:51-59 + shortcut :36-48; ReLU after the add (:70)."""
    sc = x
    l = F.relu(_bn_apply(F.conv2d(x, w[f"{p}/conv1/W"]), w[f"{p}/conv1/bn"]))
    if stride == 2:
        l = F.pad(l, (0, 1, 0, 1))
        l = F.conv2d(l, w[f"{p}/conv2/W"], stride=2)
    else:
        l = F.conv2d(l, w[f"{p}/conv2/W"], padding=1)
    l = F.relu(_bn_apply(l, w[f"{p}/conv2/bn"]))
    l = _bn_apply(F.conv2d(l, w[f"{p}/conv3/W"]), w[f"{p}/conv3/bn"])
    if f"{p}/convshortcut/W" in w:
        if stride == 2:
            sc = sc[:, :, :-1, :-1]
        sc = _bn_apply(F.conv2d(sc, w[f"{p}/convshortcut/W"], stride=stride), w[f"{p}/convshortcut/bn"])
    return F.relu(l + sc)


def _group(w, x, g, feat, count, stride):
    for i in range(count):
        x = _bottleneck(w, f"group{g}/block{i}", x, feat, stride if i == 0 else 1)
    return x


def resnet_conv4(w, x, num_blocks=RESNET_NUM_BLOCK):
    """basemodel.py:74-89."""
    l = F.pad(x, (2, 3, 2, 3))
    l = F.relu(_bn_apply(F.conv2d(l, w["conv0/W"], stride=2), w["conv0/bn"]))
    l = F.pad(l, (0, 1, 0, 1))
    l = F.max_pool2d(l, 3, stride=2)
    l = _group(w, l, 0, 64, num_blocks[0], 1)
    l = _group(w, l, 1, 128, num_blocks[1], 2)
    l = _group(w, l, 2, 256, num_blocks[2], 2)
    return l


def resnet_conv5(w, x, num_block=RESNET_NUM_BLOCK[3]):
    return _group(w, x, 3, 512, num_block, 2)


def rpn_head(w, fm):
    """model.py:30-51 -> label_logits [fH,fW,NA], box_logits [fH,fW,NA,4]."""
    hidden = F.relu(F.conv2d(fm, w["rpn/conv0/W"], w["rpn/conv0/b"], padding=1))
    lab = F.conv2d(hidden, w["rpn/class/W"], w["rpn/class/b"])[0].permute(1, 2, 0)
    box = F.conv2d(hidden, w["rpn/box/W"], w["rpn/box/b"])[0].permute(1, 2, 0)
    return lab.contiguous(), box.reshape(box.shape[0], box.shape[1], NUM_ANCHOR, 4).contiguous()


# ---------------------------------------------------------------------------------------------------
# boxes (numpy float32, element-wise op order as in the TF graph)
def decode_bbox_target(pred: np.ndarray, anchors: np.ndarray) -> np.ndarray:
    """model.py:113-139 on [...,4] float32."""
    pred, anchors = pred.astype(np.float32), anchors.astype(np.float32)
    txty, twth = pred[..., :2], pred[..., 2:]
    a1, a2 = anchors[..., :2], anchors[..., 2:]
    waha = a2 - a1
    xaya = (a2 + a1) * np.float32(0.5)
    wbhb = np.exp(np.minimum(twth, BBOX_DECODE_CLIP)).astype(np.float32) * waha
    xbyb = txty * waha + xaya
    return np.concatenate([xbyb - wbhb * np.float32(0.5), xbyb + wbhb * np.float32(0.5)], -1).astype(np.float32)


def clip_boxes(boxes: np.ndarray, h: int, w: int) -> np.ndarray:
    """model.py:17-27: max(boxes,0) then min with [w,h,w,h]."""
    b = np.maximum(boxes, np.float32(0))
    return np.minimum(b, np.array([w, h, w, h], np.float32))


def iou_tf(a: np.ndarray, b: np.ndarray) -> np.float32:
    """TF non_max_suppression_op IoU (corner order free: min/max per axis; 0 when an area <= 0)."""
    f = np.float32
    y0a, y1a = min(a[0], a[2]), max(a[0], a[2])
    x0a, x1a = min(a[1], a[3]), max(a[1], a[3])
    y0b, y1b = min(b[0], b[2]), max(b[0], b[2])
    x0b, x1b = min(b[1], b[3]), max(b[1], b[3])
    area_a = f(f(y1a - y0a) * f(x1a - x0a))
    area_b = f(f(y1b - y0b) * f(x1b - x0b))
    if area_a <= 0 or area_b <= 0:
        return f(0)
    ih = max(f(min(y1a, y1b) - max(y0a, y0b)), f(0))
    iw = max(f(min(x1a, x1b) - max(x0a, x0b)), f(0))
    inter = f(ih * iw)
    return f(inter / f(f(area_a + area_b) - inter))


def nms_tf(boxes: np.ndarray, scores: np.ndarray, max_out: int, thresh: float) -> List[int]:
    """tf.image.non_max_suppression: visit by descending score (ties: lower index), keep unless IoU with an
    already kept box is > thresh.  Returns indices into ``boxes`` in selection order."""
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    keep: List[int] = []
    t = np.float32(thresh)
    for i in order:
        if len(keep) >= max_out:
            break
        if all(not (iou_tf(boxes[i], boxes[j]) > t) for j in reversed(keep)):
            keep.append(i)
    return keep


def topk_indices(scores: np.ndarray, k: int) -> np.ndarray:
    """The SET tf.nn.top_k selects (ties toward the lower index), returned sorted by (-score, index)."""
    order = np.lexsort((np.arange(len(scores)), -scores.astype(np.float64)))
    return order[:k]


def generate_rpn_proposals(boxes: np.ndarray, scores: np.ndarray, h: int, w: int):
    """model.py:169-217 (inference).  Returns (boxes [k,4], scores [k], indices into the flat anchor list)."""
    k = min(TEST_PRE_NMS_TOPK, scores.size)
    idx = topk_indices(scores, k)
    tb = clip_boxes(boxes[idx], h, w)
    wh = tb[:, 2:] - tb[:, :2]
    valid = np.all(wh > RPN_MIN_SIZE, axis=1)
    tb, ts, idx = tb[valid], scores[idx][valid], idx[valid]
    keep = nms_tf(tb[:, [1, 0, 3, 2]], ts, TEST_POST_NMS_TOPK, RPN_PROPOSAL_NMS_THRESH)
    return tb[keep], ts[keep], idx[keep]


# ---------------------------------------------------------------------------------------------------
def roi_align(fm: torch.Tensor, boxes: np.ndarray, out: int = 14) -> torch.Tensor:
    """model.py:300-374: crop_and_resize to 2*out with the fpcoor box remap, then 2x2 average.
    fm [1,C,H,W]; boxes [N,4] x1y1x2y2 already in feature-map coordinates."""
    f = np.float32
    _, c, H, W = fm.shape
    crop = 2 * out
    n = boxes.shape[0]
    res = torch.zeros((n, c, crop, crop), dtype=torch.float32)
    fmn = fm[0]
    for r in range(n):
        x0, y0, x1, y1 = (f(v) for v in boxes[r])
        sw, sh = f(f(x1 - x0) / f(crop)), f(f(y1 - y0) / f(crop))
        nx0 = f(f(f(x0 + f(sw / f(2))) - f(0.5)) / f(W - 1))
        ny0 = f(f(f(y0 + f(sh / f(2))) - f(0.5)) / f(H - 1))
        nw = f(f(sw * f(crop - 1)) / f(W - 1))
        nh = f(f(sh * f(crop - 1)) / f(H - 1))
        by1, bx1, by2, bx2 = ny0, nx0, f(ny0 + nh), f(nx0 + nw)
        hs = f(f(f(by2 - by1) * f(H - 1)) / f(crop - 1))
        ws = f(f(f(bx2 - bx1) * f(W - 1)) / f(crop - 1))
        ys = (f(by1 * f(H - 1)) + np.arange(crop, dtype=np.float32) * hs).astype(np.float32)
        xs = (f(bx1 * f(W - 1)) + np.arange(crop, dtype=np.float32) * ws).astype(np.float32)
        yok = (ys >= 0) & (ys <= H - 1)
        xok = (xs >= 0) & (xs <= W - 1)
        yt = np.clip(np.floor(ys), 0, H - 1).astype(np.int64)
        yb = np.clip(np.ceil(ys), 0, H - 1).astype(np.int64)
        xl = np.clip(np.floor(xs), 0, W - 1).astype(np.int64)
        xr = np.clip(np.ceil(xs), 0, W - 1).astype(np.int64)
        yl = torch.from_numpy((ys - np.floor(ys)).astype(np.float32))[None, :, None]
        xlp = torch.from_numpy((xs - np.floor(xs)).astype(np.float32))[None, None, :]
        tl = fmn[:, yt][:, :, xl]
        tr = fmn[:, yt][:, :, xr]
        bl = fmn[:, yb][:, :, xl]
        br = fmn[:, yb][:, :, xr]
        top = tl + (tr - tl) * xlp
        bot = bl + (br - bl) * xlp
        val = top + (bot - top) * yl
        m = torch.from_numpy(yok[:, None] & xok[None, :])
        res[r] = val * m
    return F.avg_pool2d(res, 2, 2)


def fastrcnn_heads(w, feat: torch.Tensor):
    """model.py:377-395, 551-565: GAP + FCs.  feat [N,2048,7,7]."""
    g = feat.mean(dim=(2, 3))
    cls = g @ w["fastrcnn/class/W"].t() + w["fastrcnn/class/b"]
    box = (g @ w["fastrcnn/box/W"].t() + w["fastrcnn/box/b"]).reshape(-1, NUM_CLASS - 1, 4)
    second = g @ w["secondclassification/class/W"].t() + w["secondclassification/class/b"]
    return cls, box, second


def fastrcnn_tail(cls_logits: np.ndarray, box_logits: np.ndarray, proposals: np.ndarray, h: int, w: int):
    """train.py:275-295 + model.py:438-491 for NUM_CLASS=2.
    Returns final_boxes [M,4], final_probs [M], final_labels [M], proposal indices [M]; ordered by
    descending prob (ties lower index) -- TF leaves the order of top_k(sorted=False) unspecified."""
    if cls_logits.shape[0] == 0:
        return (np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), np.zeros((0,), np.int64),
                np.zeros((0,), np.int64))
    z = cls_logits.astype(np.float32)
    e = np.exp(z - z.max(1, keepdims=True)).astype(np.float32)
    probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
    anchors = proposals[:, None, :].repeat(NUM_CLASS - 1, 1)
    dec = decode_bbox_target(box_logits.astype(np.float32) / FASTRCNN_BBOX_REG_WEIGHTS, anchors)
    dec = clip_boxes(dec, h, w)[:, 0]
    p = probs[:, 1]
    ids = np.nonzero(p > np.float32(RESULT_SCORE_THRESH))[0]
    sel = nms_tf(dec[ids], p[ids], RESULTS_PER_IM, FASTRCNN_NMS_THRESH)
    sel = ids[sel]
    order = sorted(sel.tolist(), key=lambda i: (-float(p[i]), i))[:RESULTS_PER_IM]
    order = np.array(order, np.int64)
    return dec[order], p[order], np.ones(len(order), np.int64), order


def maskrcnn_masks(w, fm: torch.Tensor, final_boxes: np.ndarray, num_block=RESNET_NUM_BLOCK[3]) -> np.ndarray:
    """train.py:297-309 + model.py:494-509 (MODE_MASK): RoIAlign on the FINAL boxes -> conv5 -> deconv 2x2 s2 + ReLU
    -> 1x1 -> sigmoid.  Returns [M,14,14] float32 (the single foreground category)."""
    if final_boxes.shape[0] == 0:
        return np.zeros((0, 14, 14), np.float32)
    roi = roi_align(fm, final_boxes * np.float32(1.0 / ANCHOR_STRIDE), 14)
    f5 = resnet_conv5(w, roi, num_block)
    l = F.relu(F.conv_transpose2d(f5, w["maskrcnn/deconv/W"], w["maskrcnn/deconv/b"], stride=2))
    l = F.conv2d(l, w["maskrcnn/conv/W"], w["maskrcnn/conv/b"])
    return torch.sigmoid(l[:, 0]).numpy()


def fill_full_mask(box: np.ndarray, mask: np.ndarray, shape) -> np.ndarray:
    """eval.py:35-58: paste a 14x14 mask into the frame (cv2.resize float INTER_LINEAR, > 0.5)."""
    from . import cv_resize_oracle as R
    x0, y0 = list(map(int, box[:2] + 0.5))
    x1, y1 = list(map(int, box[2:] - 0.5))
    x1, y1 = max(x0, x1), max(y0, y1)
    wd, ht = x1 + 1 - x0, y1 + 1 - y0
    m = (R.resize_linear_f32(np.ascontiguousarray(mask, np.float32), wd, ht) > 0.5).astype("uint8")
    ret = np.zeros(shape, dtype="uint8")
    ret[y0:y1 + 1, x0:x1 + 1] = m[:max(0, min(ht, shape[0] - y0)), :max(0, min(wd, shape[1] - x0))]
    return ret


def model_forward(w, resized_img_bgr: np.ndarray, num_blocks=RESNET_NUM_BLOCK, intermediates=False):
    """The TF graph of Model._build_graph (inference): uint8/float HWC BGR image -> final boxes/probs/labels."""
    x = image_preprocess(torch.from_numpy(np.ascontiguousarray(resized_img_bgr)))
    h, wd = x.shape[2:]
    with torch.no_grad():
        fm = resnet_conv4(w, x, num_blocks[:3])
        lab, box = rpn_head(w, fm)
        fh, fw = lab.shape[:2]
        anchors = all_anchors(fh, fw)
        dec = decode_bbox_target(box.numpy(), anchors).reshape(-1, 4)
        pb, ps, pidx = generate_rpn_proposals(dec, lab.numpy().reshape(-1), h, wd)
        inter = {"featuremap": fm, "rpn_logits": lab, "rpn_box": box, "proposals": pb, "proposal_scores": ps,
                 "proposal_idx": pidx}
        if pb.shape[0] == 0:
            out = fastrcnn_tail(np.zeros((0, 2)), np.zeros((0, 1, 4)), pb, h, wd)
        else:
            roi = roi_align(fm, pb * np.float32(1.0 / ANCHOR_STRIDE), 14)
            f5 = resnet_conv5(w, roi, num_blocks[3])
            cls, bx, second = fastrcnn_heads(w, f5)
            inter.update({"roi": roi, "feat5": f5, "cls": cls, "box": bx, "second": second})
            out = fastrcnn_tail(cls.numpy(), bx.numpy(), pb, h, wd)
    return (out, inter) if intermediates else out


# ---------------------------------------------------------------------------------------------------
# host side
def custom_resize_shape(h: int, w: int, size=SHORT_EDGE_SIZE, max_size=MAX_SIZE) -> Tuple[int, int]:
    """common.py:47-62."""
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def detect_one_image(w, img_bgr: np.ndarray, num_blocks=RESNET_NUM_BLOCK):
    """eval.py:61-110 (no masks): resize, run, un-scale, clip.  Returns boxes [M,4] (float64 like numpy does), probs."""
    from . import cv_resize_oracle as R
    h, wd = img_bgr.shape[:2]
    newh, neww = custom_resize_shape(h, wd)
    resized = R.resize_linear_u8(img_bgr, neww, newh)
    scale = (resized.shape[0] * 1.0 / h + resized.shape[1] * 1.0 / wd) / 2
    boxes, probs, labels, _ = model_forward(w, resized, num_blocks)
    boxes = boxes / scale
    boxes = boxes.reshape(-1, 4)
    boxes[:, [0, 1]] = np.maximum(boxes[:, [0, 1]], 0)
    boxes[:, 2] = np.minimum(boxes[:, 2], wd)
    boxes[:, 3] = np.minimum(boxes[:, 3], h)
    return boxes, probs, labels


def results_to_json(boxes: np.ndarray, probs: np.ndarray) -> List[dict]:
    """train.py:388-428: xyxy -> xywh, round(x,1), round(score,2)."""
    out = []
    for b, s in zip(boxes, probs):
        b = b.copy()
        b[2] -= b[0]
        b[3] -= b[1]
        out.append({"bbox": [float(round(x, 1)) for x in b], "score": float(round(s, 2))})
    return out
