"""CPU oracle for the refinement_net forward path (DeepLabv3+ / Xception-65 on 385x385 box crops).
TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Plain PyTorch/numpy fp32 restatement of (paths relative to code/refinement_net/):

* per-box input pipeline ......... datasets/few_shot_segmentation/DAVISFewShotSegmentationDataset.py:47-70,
                                   datasets/Dataset.py:48-56,141-186, datasets/util/BoundingBox.py:15-19,
                                   datasets/Resize.py:150-193, datasets/util/Util.py:23-30, util/Normalization.py:9-37
* DeepLab wrapper ................ network/deeplab/DeepLabV3Plus.py:6-39, core/feature_extractor.py:114-116
* Xception-65 .................... network/deeplab/core/xception.py:70-293 (module), :300-345 (stride->atrous),
                                   :430-433 (stem), :496-560 (blocks), eps 1e-3 core/feature_extractor.py:202
* ASPP ........................... network/deeplab/model.py:328-435, :664-707 (split separable conv), eps 1e-5
* decoder + logits ............... network/deeplab/model.py:438-661, scale_dimension :184-197
* output layer (eval branch) ..... network/SegmentationOutputLayers.py:35-61,106-135
* per-box result ................. forwarding/FewShotSegmentationForwarder.py:85-155

Third-party arithmetic (TensorFlow 1.8 / tf.contrib.slim, absent from /root/reference and from this image)
is restated from its published semantics:
  slim.separable_conv2d(num_outputs=None): depthwise -> BN -> activation;  slim.conv2d: conv -> BN -> activation
  (no bias when a normalizer is set); conv2d_same / fixed_padding: explicit pad (k_eff-1)//2 before a VALID
  strided conv; tf.image.resize_bilinear(align_corners=False) [TF1 legacy]: src = dst*in/out, lower=floor,
  upper=min(ceil, in-1); align_corners=True: src = dst*(in-1)/(out-1); resize_nearest_neighbor legacy:
  src = min(floor(dst*in/out), in-1); tf.round = half to even.
pycocotools (absent) RLE is restated from the COCO mask API spec: column-major runs starting with zeros,
counts as delta-coded 5-bit groups + 48.

PARITY: the reference ships no test or golden vector for this path.  PINNED since round 2 by fixtures produced by EXECUTING
the reference's own python (network/deeplab/{model,common}.py, core/{xception,feature_extractor}.py with DeepLabV3Plus.py's
ModelOptions, network/SegmentationOutputLayers.py's eval branch, datasets/util/BoundingBox.py; tools/make_golden_deeplab.py on
tools/slimshim.py, an eager stand-in for TF 1.8 + slim): block table, per-layer geometry, variable names, one whole pass of
multi_scale_logits, frame-size mask / posterior of SegmentationSoftmax (tests/test_cpu_refinement_ref.py; HIP twin
tests/test_gpu_refinement_ref.py).  slim conv / batch norm / resize primitives remain restated; COCO RLE (pycocotools, absent)
is pinned only by its public spec and round trips.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

INPUT_SIZE = 385                       # configs/run:29
MARGIN = 50                            # Resize.py:151
IMAGENET_RGB_MEAN = np.array([0.485, 0.456, 0.406], np.float32)
IMAGENET_RGB_STD = np.array([0.229, 0.224, 0.225], np.float32)
EPS_BACKBONE, EPS_HEAD = 1e-3, 1e-5
ATROUS_RATES = (6, 12, 18)
# (scope, depth_list, skip, relu_inside, units, stride)   xception.py:506-551
BLOCKS = (
    ("entry_flow/block1", (128, 128, 128), "conv", False, 1, 2),
    ("entry_flow/block2", (256, 256, 256), "conv", False, 1, 2),
    ("entry_flow/block3", (728, 728, 728), "conv", False, 1, 2),
    ("middle_flow/block1", (728, 728, 728), "sum", False, 16, 1),
    ("exit_flow/block1", (728, 1024, 1024), "conv", False, 1, 2),
    ("exit_flow/block2", (1536, 1536, 2048), "none", True, 1, 1),
)
DECODER_SKIP = "entry_flow/block2/unit_1/xception_module/separable_conv2_pointwise"


def scale_dimension(dim: int, scale: float) -> int:
    return int((float(dim) - 1.0) * scale + 1.0)


# ---------------------------------------------------------------------------------------------------
# synthetic weights
def _g(name, seed):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) + 15485863 * seed) & 0x7FFFFFFF)


def _bn(name, c, seed, gamma=(0.8, 1.2)):
    g = _g(name + "/BatchNorm", seed)
    return {"gamma": torch.rand(c, generator=g) * (gamma[1] - gamma[0]) + gamma[0],
            "beta": torch.randn(c, generator=g) * 0.1,
            "mean": torch.randn(c, generator=g) * 0.1,
            "var": torch.rand(c, generator=g) + 0.5}


def _conv(name, cout, cin, k, seed, gain=2.0):
    return torch.randn((cout, cin, k, k), generator=_g(name, seed)) * math.sqrt(gain / (cin * k * k))


def _dw(name, c, seed, gain=2.0):
    return torch.randn((c, 1, 3, 3), generator=_g(name, seed)) * math.sqrt(gain / 9.0)


def plan_modules(num_middle: int = 16):
    """Flat description of every xception module actually built for output_stride 16: yields
    (prefix, cin, depth_list, skip, relu_inside, stride, rate)."""
    mods = []
    cin, current_stride, rate = 64, 1, 1
    for scope, depths, skip, relu_in, units, stride in BLOCKS:
        units = num_middle if scope.startswith("middle") else units
        for u in range(units):
            if current_stride == 8:                 # xception.py:336-341 (output_stride/2 == 8)
                mods.append((f"{scope}/unit_{u + 1}/xception_module", cin, depths, skip, relu_in, 1, rate))
                rate *= stride
            else:
                mods.append((f"{scope}/unit_{u + 1}/xception_module", cin, depths, skip, relu_in, stride, 1))
                current_stride *= stride
            cin = depths[-1]
    return mods


def synth_weights(seed: int = 0, num_middle: int = 16) -> Dict[str, object]:
    """Keys follow the slim variable scopes without the 'xception_65/' prefix for the
    backbone: '<scope>/weights' (OIHW here), '<scope>/depthwise_weights' ([C,1,3,3]), '<scope>/BatchNorm' dict."""
    w: Dict[str, object] = {}
    w["entry_flow/conv1_1/weights"] = _conv("c11", 32, 4, 3, seed)
    w["entry_flow/conv1_1/BatchNorm"] = _bn("c11", 32, seed)
    w["entry_flow/conv1_2/weights"] = _conv("c12", 64, 32, 3, seed)
    w["entry_flow/conv1_2/BatchNorm"] = _bn("c12", 64, seed)
    for prefix, cin, depths, skip, relu_in, stride, rate in plan_modules(num_middle):
        c = cin
        for i, d in enumerate(depths):
            s = f"{prefix}/separable_conv{i + 1}"
            w[s + "_depthwise/depthwise_weights"] = _dw(s + "dw", c, seed, gain=2.0 if relu_in else 1.0)
            w[s + "_depthwise/BatchNorm"] = _bn(s + "dw", c, seed)
            last = (i == 2 and skip != "none")
            w[s + "_pointwise/weights"] = _conv(s + "pw", d, c, 1, seed, gain=2.0 if (relu_in or i < 2) else 1.0)
            w[s + "_pointwise/BatchNorm"] = _bn(s + "pw", d, seed, gamma=(0.15, 0.35) if last else (0.8, 1.2))
            c = d
        if skip == "conv":
            w[prefix + "/shortcut/weights"] = _conv(prefix + "sc", depths[-1], cin, 1, seed, gain=1.0)
            w[prefix + "/shortcut/BatchNorm"] = _bn(prefix + "sc", depths[-1], seed)
    w["image_pooling/weights"] = _conv("ip", 256, 2048, 1, seed)
    w["image_pooling/BatchNorm"] = _bn("ip", 256, seed)
    w["aspp0/weights"] = _conv("aspp0", 256, 2048, 1, seed)
    w["aspp0/BatchNorm"] = _bn("aspp0", 256, seed)
    for i in (1, 2, 3):
        w[f"aspp{i}_depthwise/depthwise_weights"] = _dw(f"aspp{i}dw", 2048, seed)
        w[f"aspp{i}_depthwise/BatchNorm"] = _bn(f"aspp{i}dw", 2048, seed)
        w[f"aspp{i}_pointwise/weights"] = _conv(f"aspp{i}pw", 256, 2048, 1, seed)
        w[f"aspp{i}_pointwise/BatchNorm"] = _bn(f"aspp{i}pw", 256, seed)
    w["concat_projection/weights"] = _conv("cp", 256, 1280, 1, seed)
    w["concat_projection/BatchNorm"] = _bn("cp", 256, seed)
    w["decoder/feature_projection0/weights"] = _conv("fp0", 48, 256, 1, seed)
    w["decoder/feature_projection0/BatchNorm"] = _bn("fp0", 48, seed)
    for j, cin in ((0, 304), (1, 256)):
        w[f"decoder/decoder_conv{j}_depthwise/depthwise_weights"] = _dw(f"dec{j}dw", cin, seed)
        w[f"decoder/decoder_conv{j}_depthwise/BatchNorm"] = _bn(f"dec{j}dw", cin, seed)
        w[f"decoder/decoder_conv{j}_pointwise/weights"] = _conv(f"dec{j}pw", 256, cin, 1, seed)
        w[f"decoder/decoder_conv{j}_pointwise/BatchNorm"] = _bn(f"dec{j}pw", 256, seed)
    w["logits/features/weights"] = _conv("logits", 2, 256, 1, seed, gain=20.0)
    w["logits/features/biases"] = torch.randn(2, generator=_g("logitsb", seed)) * 0.5
    return w


# ---------------------------------------------------------------------------------------------------
# TF1 resize semantics
def _lerp_idx(out: int, inn: int, align_corners: bool):
    if align_corners and out > 1:
        scale = np.float32(inn - 1) / np.float32(out - 1)
    else:
        scale = np.float32(inn) / np.float32(out)
    src = (np.arange(out, dtype=np.float32) * scale).astype(np.float32)
    lo = np.floor(src).astype(np.int64)
    hi = np.minimum(np.ceil(src).astype(np.int64), inn - 1)
    return lo, hi, (src - lo.astype(np.float32)).astype(np.float32)


def resize_bilinear_tf(x: torch.Tensor, oh: int, ow: int, align_corners: bool = False) -> torch.Tensor:
    """tf.image.resize_bilinear (TF1, no half-pixel centres) on [N,C,H,W]."""
    _, _, h, w = x.shape
    ylo, yhi, yl = _lerp_idx(oh, h, align_corners)
    xlo, xhi, xl = _lerp_idx(ow, w, align_corners)
    yl = torch.from_numpy(yl)[None, None, :, None]
    xl = torch.from_numpy(xl)[None, None, None, :]
    tl = x[:, :, ylo][:, :, :, xlo]
    tr = x[:, :, ylo][:, :, :, xhi]
    bl = x[:, :, yhi][:, :, :, xlo]
    br = x[:, :, yhi][:, :, :, xhi]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return top + (bot - top) * yl


def resize_nearest_tf(x: torch.Tensor, oh: int, ow: int) -> torch.Tensor:
    """tf.image.resize_nearest_neighbor (TF1 legacy, align_corners=False) on [N,C,H,W]."""
    _, _, h, w = x.shape
    ys = np.minimum(np.floor(np.arange(oh, dtype=np.float32) * (np.float32(h) / np.float32(oh))).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(ow, dtype=np.float32) * (np.float32(w) / np.float32(ow))).astype(np.int64), w - 1)
    return x[:, :, ys][:, :, :, xs]


# ---------------------------------------------------------------------------------------------------
# input pipeline
def crop_box(bbox_y0x0y1x1, h: int, w: int) -> Tuple[int, int, int, int]:
    """Resize.py:156-166: tf.round (half to even), +-MARGIN, clip."""
    y0, x0, y1, x1 = (int(v) for v in np.round(np.asarray(bbox_y0x0y1x1, np.float32)))
    return max(y0 - MARGIN, 0), max(x0 - MARGIN, 0), min(y1 + MARGIN, h), min(x1 + MARGIN, w)


def make_input(img_u8_rgb: np.ndarray, bbox_y0x0y1x1) -> Tuple[torch.Tensor, Tuple[int, int, int, int]]:
    """One proposal -> network input [1,4,385,385] (normalised RGB + {0,1} guidance) and the crop box."""
    h, w = img_u8_rgb.shape[:2]
    img = img_u8_rgb.astype("float32") / 255                      # DAVISFewShotSegmentationDataset.py:51
    guid = np.zeros((h, w), np.float32)
    gy0, gx0, gy1, gx1 = (int(v) for v in np.round(np.asarray(bbox_y0x0y1x1, np.float32)))   # BoundingBox.py:15-19
    guid[max(gy0, 0):max(gy1, 0), max(gx0, 0):max(gx1, 0)] = 1
    y0, x0, y1, x1 = crop_box(bbox_y0x0y1x1, h, w)
    ci = torch.from_numpy(img[y0:y1, x0:x1]).permute(2, 0, 1)[None]
    cg = torch.from_numpy(guid[y0:y1, x0:x1])[None, None]
    ri = resize_bilinear_tf(ci, INPUT_SIZE, INPUT_SIZE, False)
    rg = resize_nearest_tf(cg, INPUT_SIZE, INPUT_SIZE)
    ri = (ri - torch.from_numpy(IMAGENET_RGB_MEAN).view(1, 3, 1, 1)) / torch.from_numpy(IMAGENET_RGB_STD).view(1, 3, 1, 1)
    return torch.cat([ri, rg], 1), (y0, x0, y1, x1)


def deeplab_preprocess(x: torch.Tensor) -> torch.Tensor:
    """DeepLabV3Plus.py:12-14 unnormalize*255, then feature_extractor.py:114-116 (2/255)x - 1."""
    mean = torch.cat([torch.from_numpy(IMAGENET_RGB_MEAN), torch.zeros(1)]).view(1, 4, 1, 1)
    std = torch.cat([torch.from_numpy(IMAGENET_RGB_STD), torch.ones(1)]).view(1, 4, 1, 1)
    x = (x * std + mean) * 255
    return np.float32(2.0 / 255.0) * x - 1.0


# ---------------------------------------------------------------------------------------------------
# network
def _bn_apply(x, bn, eps):
    return F.batch_norm(x, bn["mean"], bn["var"], bn["gamma"], bn["beta"], training=False, eps=eps)


def _conv_same(x, wt, stride):
    """resnet_utils.conv2d_same (3x3)."""
    if stride == 1:
        return F.conv2d(x, wt, padding=1)
    return F.conv2d(F.pad(x, (1, 1, 1, 1)), wt, stride=stride)


def _sep_conv(w, scope, x, stride, rate, relu_inside, eps):
    """separable_conv2d_same split form (xception.py:92-206): dw -> BN [-> ReLU] -> pw -> BN [-> ReLU]."""
    c = x.shape[1]
    dw = w[scope + "_depthwise/depthwise_weights"]
    if stride == 1:
        y = F.conv2d(x, dw, padding=rate, dilation=rate, groups=c)
    else:
        p = rate                                    # fixed_padding: k_eff = 3 + 2(rate-1); pad (k_eff-1)//2 each side
        y = F.conv2d(F.pad(x, (p, p, p, p)), dw, stride=stride, dilation=rate, groups=c)
    y = _bn_apply(y, w[scope + "_depthwise/BatchNorm"], eps)
    if relu_inside:
        y = F.relu(y)
    y = _bn_apply(F.conv2d(y, w[scope + "_pointwise/weights"]), w[scope + "_pointwise/BatchNorm"], eps)
    if relu_inside:
        y = F.relu(y)
    return y


def xception_65(w, x: torch.Tensor, num_middle: int = 16, end_points=None):
    x = F.relu(_bn_apply(_conv_same(x, w["entry_flow/conv1_1/weights"], 2), w["entry_flow/conv1_1/BatchNorm"], EPS_BACKBONE))
    x = F.relu(_bn_apply(_conv_same(x, w["entry_flow/conv1_2/weights"], 1), w["entry_flow/conv1_2/BatchNorm"], EPS_BACKBONE))
    for prefix, cin, depths, skip, relu_in, stride, rate in plan_modules(num_middle):
        inp, res = x, x
        for i in range(3):
            if not relu_in:
                res = F.relu(res)
            res = _sep_conv(w, f"{prefix}/separable_conv{i + 1}", res, stride if i == 2 else 1, rate, relu_in, EPS_BACKBONE)
            if end_points is not None:
                end_points[f"{prefix}/separable_conv{i + 1}_pointwise"] = res
        if skip == "conv":
            sc = _bn_apply(F.conv2d(inp, w[prefix + "/shortcut/weights"], stride=stride), w[prefix + "/shortcut/BatchNorm"],
                           EPS_BACKBONE)
            x = res + sc
        elif skip == "sum":
            x = res + inp
        else:
            x = res
    return x


def deeplab_logits(w, net_in: torch.Tensor, num_middle: int = 16, intermediates=None) -> torch.Tensor:
    """[N,4,385,385] normalised input -> logits [N,2,97,97] (model.py:200-325 with one scale)."""
    x = deeplab_preprocess(net_in)
    ep: Dict[str, torch.Tensor] = {}
    feat = xception_65(w, x, num_middle, ep)
    n, _, fh, fw = feat.shape
    act = lambda t, name: F.relu(_bn_apply(t, w[name + "/BatchNorm"], EPS_HEAD))   # noqa: E731
    img = feat.mean(dim=(2, 3), keepdim=True)
    img = act(F.conv2d(img, w["image_pooling/weights"]), "image_pooling")
    branches = [resize_bilinear_tf(img, fh, fw, True), act(F.conv2d(feat, w["aspp0/weights"]), "aspp0")]
    for i, r in enumerate(ATROUS_RATES, 1):
        y = act(F.conv2d(feat, w[f"aspp{i}_depthwise/depthwise_weights"], padding=r, dilation=r, groups=feat.shape[1]),
                f"aspp{i}_depthwise")
        branches.append(act(F.conv2d(y, w[f"aspp{i}_pointwise/weights"]), f"aspp{i}_pointwise"))
    y = act(F.conv2d(torch.cat(branches, 1), w["concat_projection/weights"]), "concat_projection")
    dh = scale_dimension(net_in.shape[2], 0.25)
    dw_ = scale_dimension(net_in.shape[3], 0.25)
    skip = act(F.conv2d(ep[DECODER_SKIP], w["decoder/feature_projection0/weights"]), "decoder/feature_projection0")
    d = torch.cat([resize_bilinear_tf(y, dh, dw_, True), resize_bilinear_tf(skip, dh, dw_, True)], 1)
    for j in (0, 1):
        d = act(F.conv2d(d, w[f"decoder/decoder_conv{j}_depthwise/depthwise_weights"], padding=1, groups=d.shape[1]),
                f"decoder/decoder_conv{j}_depthwise")
        d = act(F.conv2d(d, w[f"decoder/decoder_conv{j}_pointwise/weights"]), f"decoder/decoder_conv{j}_pointwise")
    logits = F.conv2d(d, w["logits/features/weights"], w["logits/features/biases"])
    logits = resize_bilinear_tf(logits, dh, dw_, True)             # model.py:295-297 (identity size)
    if intermediates is not None:
        intermediates.update({"xception": feat, "aspp": y, "decoder": d, "skip": ep[DECODER_SKIP]})
    return logits


def output_layer(logits: torch.Tensor, crop: Tuple[int, int, int, int], h: int, w: int):
    """SegmentationOutputLayers.py:35-61,106-135 eval branch for ONE box: logits [1,2,97,97] ->
    (mask [h,w] uint8 in {0,1}, posterior [h,w] float32)."""
    lg = resize_bilinear_tf(logits, INPUT_SIZE, INPUT_SIZE, False)
    prob = torch.softmax(lg, dim=1)[:, 1:2]
    pred = lg.argmax(dim=1, keepdim=True).to(torch.float32)
    y0, x0, y1, x1 = crop
    hc, wc = y1 - y0, x1 - x0
    mask = torch.zeros((h, w), dtype=torch.uint8)
    post = torch.zeros((h, w), dtype=torch.float32)
    if hc > 0 and wc > 0:
        mask[y0:y1, x0:x1] = resize_nearest_tf(pred, hc, wc)[0, 0].to(torch.uint8)
        post[y0:y1, x0:x1] = resize_bilinear_tf(prob, hc, wc, False)[0, 0]
    return mask.numpy(), post.numpy()


def conf_score(mask: np.ndarray, post: np.ndarray) -> np.float32:
    """FewShotSegmentationForwarder.py:144-148."""
    c = post.copy()
    c[mask == 0] = 1 - post[mask == 0]
    c = 2 * c - 1
    return c[:].mean()


# ---------------------------------------------------------------------------------------------------
# COCO RLE (pycocotools.mask.encode restated)
def rle_counts(mask: np.ndarray) -> List[int]:
    """Column-major run lengths, alternating 0s/1s, starting with the number of leading zeros."""
    flat = (np.asarray(mask) != 0).astype(np.uint8).flatten(order="F")
    if flat.size == 0:
        return []
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate([[0], change, [flat.size]])
    runs = np.diff(bounds).tolist()
    if flat[0] == 1:
        runs = [0] + runs
    return runs


def rle_to_string(counts: List[int]) -> str:
    """maskApi.c rleToString: counts[i>2] are delta-coded against counts[i-2]; 5 data bits + continuation bit, +48."""
    out = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return "".join(out)


def rle_from_string(s: str) -> List[int]:
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_encode(mask: np.ndarray) -> dict:
    h, w = mask.shape
    return {"size": [int(h), int(w)], "counts": rle_to_string(rle_counts(mask))}


def rle_decode(rle: dict) -> np.ndarray:
    h, w = rle["size"]
    counts = rle_from_string(rle["counts"])
    flat = np.zeros(h * w, np.uint8)
    pos, val = 0, 0
    for c in counts:
        flat[pos:pos + c] = val
        pos += c
        val ^= 1
    return flat.reshape((h, w), order="F")


def refine_proposals(w, img_u8_rgb: np.ndarray, proposals: List[dict], num_middle: int = 16) -> List[dict]:
    """The forwarder loop (FewShotSegmentationForwarder.py:104-149) for one frame."""
    h, wd = img_u8_rgb.shape[:2]
    out = [dict(p) for p in proposals]
    with torch.no_grad():
        for i, p in enumerate(proposals):
            x0, y0, bw, bh = p["bbox"]
            bbox = [y0, x0, y0 + bh, x0 + bw]
            net_in, crop = make_input(img_u8_rgb, bbox)
            mask, post = output_layer(deeplab_logits(w, net_in, num_middle), crop, h, wd)
            out[i]["segmentation"] = rle_encode(mask)
            out[i]["conf_score"] = str(conf_score(mask, post))
    return out
