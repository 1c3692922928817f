"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch fp32 / numpy) of the ReID embedding path.
Only tests/ may import this; the product (premvos_amd/reid) runs on HIP kernels and has no CPU fallback.

Restated from (paths relative to code/ReID_net/):
  * network ............. configs/run:37-70 (the layer table), network/NetworkLayers.py:101-210 (Conv, ResidualUnit2),
                          :231-250 (FullyConnected), network/NetworkOutputLayers.py:253-272 (the 128-d output, linear),
                          network/Util_Network.py:12-30 (SAME conv / SAME max-pool), :64-80 (NHWC flatten),
                          NetworkLayers.py:12-13,38-60 (inference BatchNorm on the moving statistics, eps 1e-5)
  * crops, in-merge ..... datasets/Similarity/DAVIS_Forward_Feed.py:36-96 (context region 1.2, tf.round, clip with an excess
                          of AT LEAST ONE pixel, zeros when min(h, w) <= 10), MergeTrack/ReID_net_functions.py:26-45
  * crops, batch stage .. datasets/Similarity/Similarity.py:264-298 (same, excess >= 0, no small-box rule),
                          DAVIS_Forward_Similarity.py:25-43 (bbox = toBbox(segmentation), w/h <= 0 skipped, tag)
  * resize / normalise .. datasets/Util/Resize.py:55-61 (tf.image.resize_images bilinear, TF1 legacy coordinates
                          src = dst*in/out), datasets/Util/Normalization.py:9-21
  * export .............. Forwarding/ReIDForwarding.py:34-92 (proposal JSON gains "ReID": list of 128 floats)

Third-party arithmetic (TensorFlow 1.x conv2d/max_pool SAME, resize_images, tf.round) is absent from this image and is
restated from its published semantics.  PARITY: the reference holds no test or golden vector for this path; PINNED since round 2
by fixtures produced by EXECUTING its own python (Config.py on configs/run, network/Network.py:build_tower instantiating
NetworkLayers.py / NetworkOutputLayers.py, datasets/Similarity/DAVIS_Forward_Feed.py's crop pipeline; tools/make_golden_reid.py
on tools/tfshim.py): layer table, every unit's wiring, variable names + shapes, whole-net activations and embedding, context
boxes and crops (tests/test_cpu_reid_ref.py; HIP twin in tests/test_gpu_reid.py).  The TF primitives remain restated.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
INPUT_SIZE = 128
CONTEXT = 1.2
IMAGENET_RGB_MEAN = np.array([0.485, 0.456, 0.406], np.float32)
IMAGENET_RGB_STD = np.array([0.229, 0.224, 0.225], np.float32)

# configs/run:37-70.  (name, n_convs, n_features per conv, filter sizes, strides per conv)
UNITS: List[Tuple[str, int, Sequence[int], Sequence[int], Sequence[int]]] = (
    [("res0", 2, (128, 128), (3, 3), (2, 1)), ("res1", 2, (128, 128), (3, 3), (1, 1)), ("res2", 2, (128, 128), (3, 3), (1, 1)),
     ("res3", 2, (256, 256), (3, 3), (2, 1)), ("res4", 2, (256, 256), (3, 3), (1, 1)), ("res5", 2, (256, 256), (3, 3), (1, 1)),
     ("res6", 2, (512, 512), (3, 3), (2, 1))]
    + [(f"res{i}", 2, (512, 512), (3, 3), (1, 1)) for i in range(7, 12)]
    + [("res12", 2, (512, 1024), (3, 3), (1, 2)), ("res13", 2, (512, 1024), (3, 3), (1, 1)),
       ("res14", 2, (512, 1024), (3, 3), (1, 1)),
       ("res15", 3, (512, 1024, 2048), (1, 3, 1), (1, 2, 1)), ("res16", 3, (1024, 2048, 4096), (1, 3, 1), (1, 1, 1))])
CONV0_FEATURES, CONV1_FEATURES, FC_FEATURES, EMBED = 64, 500, 500, 128


def unit_plan(units=UNITS, cin=CONV0_FEATURES):
    """-> [(name, cin, n_features, filters, strides, has_shortcut_conv, shortcut_stride)] following ResidualUnit2."""
    out = []
    for name, n, feats, ks, st in units:
        sres = int(np.prod(st))
        out.append((name, cin, tuple(feats), tuple(ks), tuple(st), feats[-1] != cin or sres != 1, sres))
        cin = feats[-1]
    return out


# ---------------------------------------------------------------------------------------------------
# synthetic weights, named like the TF variables (<layer>/W, <layer>/bn0/{beta,gamma,mean_ema,var_ema}, ...)
def _g(name: str, seed: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)


def _bn(name: str, c: int, seed: int) -> Dict[str, torch.Tensor]:
    g = _g(name, seed)
    return {"gamma": 1.0 + 0.1 * torch.randn(c, generator=g), "beta": 0.1 * torch.randn(c, generator=g),
            "mean": 0.1 * torch.randn(c, generator=g), "var": 0.5 + torch.rand(c, generator=g)}


def _w(name: str, cout: int, cin: int, k: int, seed: int, gain: float = 2.0) -> torch.Tensor:
    return torch.randn((cout, cin, k, k), generator=_g(name, seed)) * (gain / (cin * k * k)) ** 0.5


def final_spatial(units=UNITS, size: int = INPUT_SIZE) -> int:
    """Edge of the map that reaches fc1: SAME strides of the units, then the SAME 3x3/3 max-pool of conv1."""
    for _, _, _, _, st in units:
        for s in st:
            size = -(-size // s)
    return -(-size // 3)


def synth_weights(seed: int = 0, units=UNITS) -> Dict[str, object]:
    """OIHW conv kernels, BatchNorm dicts, FC matrices [out, in] (in = NHWC-flattened)."""
    spatial_out = final_spatial(units)
    w: Dict[str, object] = {"conv0/W": _w("conv0/W", CONV0_FEATURES, 3, 3, seed)}
    cin = CONV0_FEATURES
    for name, cin_u, feats, ks, st, short, _ in unit_plan(units):
        w[f"{name}/bn0"] = _bn(f"{name}/bn0", cin_u, seed)
        if short:
            w[f"{name}/W0"] = _w(f"{name}/W0", feats[-1], cin_u, 1, seed, gain=1.0)
        c = cin_u
        for i, (f, k) in enumerate(zip(feats, ks), 1):
            if i > 1:
                w[f"{name}/bn{i}"] = _bn(f"{name}/bn{i}", c, seed)
            w[f"{name}/W{i}"] = _w(f"{name}/W{i}", f, c, k, seed, gain=2.0 if i < len(feats) else 0.5)
            c = f
        cin = feats[-1]
    w["conv1/bn"] = _bn("conv1/bn", cin, seed)
    w["conv1/W"] = _w("conv1/W", CONV1_FEATURES, cin, 3, seed)
    nin = spatial_out * spatial_out * CONV1_FEATURES
    for name, i, o in (("fc1", nin, FC_FEATURES), ("fc2", FC_FEATURES, FC_FEATURES), ("outputTriplet", FC_FEATURES, EMBED)):
        w[f"{name}/bn"] = _bn(f"{name}/bn", i, seed)
        w[f"{name}/W"] = torch.randn((o, i), generator=_g(f"{name}/W", seed)) * (2.0 / i) ** 0.5
        w[f"{name}/b"] = 0.1 * torch.randn(o, generator=_g(f"{name}/b", seed))
    return w


# ---------------------------------------------------------------------------------------------------
# TF semantics
def same_pad(size: int, k: int, s: int) -> Tuple[int, int, int]:
    """TF 'SAME': out = ceil(size/s); total pad = max((out-1)*s + k - size, 0); the extra pixel goes AFTER."""
    out = -(-size // s)
    tot = max((out - 1) * s + k - size, 0)
    return out, tot // 2, tot - tot // 2


def conv_same(x: torch.Tensor, w: torch.Tensor, stride: int = 1) -> torch.Tensor:
    k = w.shape[2]
    _, pt, pb = same_pad(x.shape[2], k, stride)
    _, pl, pr = same_pad(x.shape[3], k, stride)
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, stride=stride)


def max_pool_same(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
    _, pt, pb = same_pad(x.shape[2], k, s)
    _, pl, pr = same_pad(x.shape[3], k, s)
    return F.max_pool2d(F.pad(x, (pl, pr, pt, pb), value=float("-inf")), k, s)


def bn(x: torch.Tensor, p: Dict[str, torch.Tensor]) -> torch.Tensor:
    shape = (1, -1, 1, 1) if x.dim() == 4 else (1, -1)
    return (x - p["mean"].view(shape)) * (p["gamma"] / torch.sqrt(p["var"] + BN_EPS)).view(shape) + p["beta"].view(shape)


def resize_bilinear_legacy(img: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """tf.image.resize_images(img, (oh, ow)) (bilinear, align_corners=False, TF1): src = dst*in/out, no half-pixel."""
    h, w = img.shape[:2]
    ys = np.arange(oh, dtype=np.float32) * np.float32(h / oh)
    xs = np.arange(ow, dtype=np.float32) * np.float32(w / ow)
    y0, x0 = np.floor(ys).astype(np.int64), np.floor(xs).astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    ty, tx = (ys - y0).astype(np.float32)[:, None, None], (xs - x0).astype(np.float32)[None, :, None]
    img = img.astype(np.float32)
    top = img[y0][:, x0] + (img[y0][:, x1] - img[y0][:, x0]) * tx
    bot = img[y1][:, x0] + (img[y1][:, x1] - img[y1][:, x0]) * tx
    return (top + (bot - top) * ty).astype(np.float32)


def context_boxes(boxes_xywh: np.ndarray, height: int, width: int, feed: bool = True) -> np.ndarray:
    """float32 arithmetic as the TF graph does it -> int32 [n,4] (x, y, w, h).  feed=True: DAVIS_Forward_Feed.py:36-60
    (excess >= 1); False: Similarity.py:267-287 (excess >= 0)."""
    b = np.asarray(boxes_xywh, np.float32).reshape(-1, 4).copy()
    xs, ys, ws, hs = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    f = np.float32(CONTEXT - 1.0)
    xs = xs - np.float32(0.5) * ws * f
    ys = ys - np.float32(0.5) * hs * f
    ws = ws * np.float32(CONTEXT)
    hs = hs * np.float32(CONTEXT)
    xs, ys, ws, hs = (np.rint(v).astype(np.int32) for v in (xs, ys, ws, hs))        # tf.round: half to even
    xs, ys = np.maximum(xs, 0), np.maximum(ys, 0)
    lo = 1 if feed else 0
    ws = ws - np.maximum(xs + ws - width, lo)
    hs = hs - np.maximum(ys + hs - height, lo)
    return np.stack([xs, ys, ws, hs], 1).astype(np.int32)


def make_crop(frame_rgb_u8: np.ndarray, box_xywh_int: Sequence[int], feed: bool = True) -> np.ndarray:
    """-> normalised float32 [128,128,3] (DAVIS_Forward_Feed.py:62-84 / Similarity.py:288-297)."""
    x, y, w, h = (int(v) for v in box_xywh_int)
    # in-merge feed: `self.image / 255` (DAVIS_Forward_Feed.py:27); batch stage: tf.image.convert_image_dtype(uint8 -> float32) of
    # load_image_tensorflow (Util/Reader.py:162) = cast * float32(1 / 255) -- the two differ in the last bit for 39 % of the values
    img = frame_rgb_u8.astype(np.float32) / np.float32(255) if feed else frame_rgb_u8.astype(np.float32) * np.float32(1.0 / 255.0)
    if feed and min(h, w) <= 10:
        out = np.zeros((INPUT_SIZE, INPUT_SIZE, 3), np.float32)
    else:
        crop = img[max(y, 0):max(y + h, 0), max(x, 0):max(x + w, 0)]             # python slicing == tf slicing here
        out = resize_bilinear_legacy(crop, INPUT_SIZE, INPUT_SIZE)
    return ((out - IMAGENET_RGB_MEAN) / IMAGENET_RGB_STD).astype(np.float32)


# ---------------------------------------------------------------------------------------------------
def forward(w: Dict[str, object], crops_nhwc: np.ndarray, units=UNITS, intermediates: Dict = None) -> np.ndarray:
    """crops [n,128,128,3] normalised -> embeddings [n,128] (the 'outputTriplet' layer, linear)."""
    x = torch.from_numpy(np.ascontiguousarray(crops_nhwc)).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        x = conv_same(x, w["conv0/W"])                                                      # activation "linear", no BN
        for name, cin, feats, ks, st, short, sres in unit_plan(units):
            a = F.relu(bn(x, w[f"{name}/bn0"]))
            res = conv_same(a, w[f"{name}/W0"], sres) if short else x
            cur = conv_same(a, w[f"{name}/W1"], st[0])
            for i in range(2, len(feats) + 1):
                cur = conv_same(F.relu(bn(cur, w[f"{name}/bn{i}"])), w[f"{name}/W{i}"], st[i - 1])
            x = cur + res
            if intermediates is not None:
                intermediates[name] = x.clone()
        x = conv_same(F.relu(bn(x, w["conv1/bn"])), w["conv1/W"])                        # BN -> relu -> conv -> pool
        x = max_pool_same(x, 3, 3)
        if intermediates is not None:
            intermediates["conv1"] = x.clone()
        h = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                                   # NHWC flatten
        for name, act in (("fc1", True), ("fc2", True), ("outputTriplet", False)):
            h = bn(h, w[f"{name}/bn"]) @ w[f"{name}/W"].t() + w[f"{name}/b"]
            if act:
                h = F.relu(h)
    return h.numpy()


def add_reid(w: Dict[str, object], frame_rgb_u8: np.ndarray, proposals: List[dict], units=UNITS) -> List[dict]:
    """MergeTrack/ReID_net_functions.py:26-45: every proposal's 'bbox' (xywh) -> 'ReID' (list of 128 floats)."""
    if not proposals:
        return proposals
    H, W = frame_rgb_u8.shape[:2]
    boxes = context_boxes(np.array([p["bbox"] for p in proposals], np.float32), H, W, feed=True)
    crops = np.stack([make_crop(frame_rgb_u8, b, feed=True) for b in boxes])
    emb = forward(w, crops, units)
    for p, e in zip(proposals, emb):
        p["ReID"] = e.tolist()
    return proposals
