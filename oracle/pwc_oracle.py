"""CPU oracle for the PWC-DC-Net flow path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``premvos_amd.flow``) runs hand-written HIP kernels and
fails loudly when the HIP library is missing.

It is a plain-PyTorch fp32 *restatement* (table driven, functional over a
``state_dict``) of the reference algorithm:

* network topology / channel bookkeeping ..... models/PWCNet.py:38-131
* forward pass ................................ models/PWCNet.py:179-272
* backward bilinear warp + validity mask ...... models/PWCNet.py:140-176
  (torch-0.2 ``grid_sample`` == bilinear / zeros / ``align_corners=True``)
* correlation (cost volume) ................... correlation_package/src/corr_cuda.c:23-45
                                                correlation_package/src/corr_cuda_kernel.cu:59-127
  PWC-Net instantiates pad=md=4, k=1, s1=s2=1, multiply  (PWCNet.py:69)

Pinning: ``tests/golden/pwc_*.npz`` were produced by *importing the reference*
``models/PWCNet.py`` in the build container (``tools/make_golden_pwc.py``,
correlation module replaced by a pure-torch stand-in because the reference's
CPU correlation entry points are stubs, corr.c:3-16).  The correlation
restatement is additionally pinned by the only known-answer the reference
holds: ``correlation(0,1,0,1,1,1)`` on [[1,2],[3,4]],[[5,6],[7,8]] ==
[[5,12],[21,32]] (test/test.py:76-77) and the 1x9x2x2 shape (:81).
"""
from __future__ import annotations

import math
import zlib

import numpy as np
import torch
import torch.nn.functional as F

MD = 4                     # max displacement            PWCNet.py:43,69
ND = (2 * MD + 1) ** 2     # 81 cost-volume channels     PWCNet.py:72
DENSE_GROWTH = (128, 128, 96, 64, 32)          # PWCNet.py:73
PYRAMID = (  # name, cin, cout, stride                    PWCNet.py:50-67
    ("conv1a", 3, 16, 2), ("conv1aa", 16, 16, 1), ("conv1b", 16, 16, 1),
    ("conv2a", 16, 32, 2), ("conv2aa", 32, 32, 1), ("conv2b", 32, 32, 1),
    ("conv3a", 32, 64, 2), ("conv3aa", 64, 64, 1), ("conv3b", 64, 64, 1),
    ("conv4a", 64, 96, 2), ("conv4aa", 96, 96, 1), ("conv4b", 96, 96, 1),
    ("conv5a", 96, 128, 2), ("conv5aa", 128, 128, 1), ("conv5b", 128, 128, 1),
    ("conv6aa", 128, 196, 2), ("conv6a", 196, 196, 1), ("conv6b", 196, 196, 1),
)
# order in which forward() chains the three convs of each pyramid level
PYRAMID_ORDER = {1: ("conv1a", "conv1aa", "conv1b"), 2: ("conv2a", "conv2aa", "conv2b"),
                 3: ("conv3a", "conv3aa", "conv3b"), 4: ("conv4a", "conv4aa", "conv4b"),
                 5: ("conv5a", "conv5aa", "conv5b"), 6: ("conv6aa", "conv6a", "conv6b")}
LEVEL_FEAT = {6: 196, 5: 128, 4: 96, 3: 64, 2: 32}
FLOW_SCALE = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}   # PWCNet.py:211,227,243,257
CONTEXT = (  # name, cin, cout, dilation                  PWCNet.py:125-131
    ("dc_conv1", None, 128, 1), ("dc_conv2", 128, 128, 2), ("dc_conv3", 128, 128, 4),
    ("dc_conv4", 128, 96, 8), ("dc_conv5", 96, 64, 16), ("dc_conv6", 64, 32, 1),
)


def level_in_channels(level: int) -> int:
    """Estimator input width 'od' per level (PWCNet.py:75,85,95,105,115)."""
    return ND if level == 6 else ND + LEVEL_FEAT[level] + 4


def param_shapes() -> "dict[str, tuple]":
    """name -> shape for every tensor of the reference state_dict (Appendix A of SURVEY)."""
    shapes = {}
    for name, cin, cout, _ in PYRAMID:
        shapes[f"{name}.0.weight"] = (cout, cin, 3, 3)
        shapes[f"{name}.0.bias"] = (cout,)
    for lvl in (6, 5, 4, 3, 2):
        od = level_in_channels(lvl)
        cin = od
        for i, g in enumerate(DENSE_GROWTH):
            shapes[f"conv{lvl}_{i}.0.weight"] = (g, cin, 3, 3)
            shapes[f"conv{lvl}_{i}.0.bias"] = (g,)
            cin += g
        shapes[f"predict_flow{lvl}.weight"] = (2, cin, 3, 3)
        shapes[f"predict_flow{lvl}.bias"] = (2,)
        shapes[f"deconv{lvl}.weight"] = (2, 2, 4, 4)          # ConvTranspose2d: [in,out,kh,kw]
        shapes[f"deconv{lvl}.bias"] = (2,)
        if lvl != 2:
            shapes[f"upfeat{lvl}.weight"] = (cin, 2, 4, 4)
            shapes[f"upfeat{lvl}.bias"] = (2,)
    cin = level_in_channels(2) + sum(DENSE_GROWTH)
    for name, c, cout, _ in CONTEXT:
        c = cin if c is None else c
        shapes[f"{name}.0.weight"] = (cout, c, 3, 3)
        shapes[f"{name}.0.bias"] = (cout,)
    shapes["dc_conv7.weight"] = (2, 32, 3, 3)
    shapes["dc_conv7.bias"] = (2,)
    return shapes


def synth_state_dict(seed: int = 0) -> "dict[str, torch.Tensor]":
    """Deterministic He-normal weights / small non-zero biases, one RNG stream per tensor.

    No checkpoint exists in the build container (simple_run.sh:12-17 downloads it), so
    parity is arithmetic parity on these weights.  Independent of construction order.
    """
    sd = {}
    for name, shape in param_shapes().items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        if name.endswith("weight"):
            is_deconv = name.startswith(("deconv", "upfeat"))
            fan_in = (shape[0] if is_deconv else shape[1]) * shape[2] * shape[3]
            if is_deconv:
                fan_in = fan_in / 4.0      # only 4 of 16 taps hit one output pixel
            std = math.sqrt(2.0 / fan_in)
            if name.startswith(("predict_flow", "dc_conv7")):
                std *= 1.0                  # flows of a few px: warps exercise in- and out-of-bounds
            sd[name] = torch.randn(shape, generator=g) * std
        else:
            sd[name] = torch.randn(shape, generator=g) * 0.05
    return sd


# --------------------------------------------------------------------------------------
# correlation
def correlation_np(f1: np.ndarray, f2: np.ndarray, pad_size=MD, kernel_size=1,
                   max_displacement=MD, stride1=1, stride2=1) -> np.ndarray:
    """Cost volume, NCHW float32 in/out.  Follows corr_cuda.c:23-45 (shape math) and
    corr_cuda_kernel.cu:59-127 (indexing; top_channel = (dy/s2+r)*D + (dx/s2+r);
    mean over kernel_size^2 * C)."""
    n, c, h, w = f1.shape
    kr = (kernel_size - 1) // 2
    border = max_displacement + kr
    ph, pw = h + 2 * pad_size, w + 2 * pad_size
    ow = int(math.ceil((pw - 2 * border) / float(stride1)))
    oh = int(math.ceil((ph - 2 * border) / float(stride1)))
    r = max_displacement // stride2
    d = 2 * r + 1
    p1 = np.zeros((n, c, ph, pw), np.float32)
    p2 = np.zeros((n, c, ph, pw), np.float32)
    p1[:, :, pad_size:pad_size + h, pad_size:pad_size + w] = f1
    p2[:, :, pad_size:pad_size + h, pad_size:pad_size + w] = f2
    out = np.zeros((n, d * d, oh, ow), np.float32)
    ys = np.arange(oh) * stride1 + max_displacement
    xs = np.arange(ow) * stride1 + max_displacement
    for tc in range(d * d):
        dx = (tc % d - r) * stride2
        dy = (tc // d - r) * stride2
        acc = np.zeros((n, oh, ow), np.float32)
        for j in range(kernel_size):
            for i in range(kernel_size):
                a = p1[:, :, ys[:, None] + j, xs[None, :] + i]
                b = p2[:, :, ys[:, None] + dy + j, xs[None, :] + dx + i]
                acc += (a * b).sum(axis=1, dtype=np.float32)
        out[:, tc] = acc / np.float32(kernel_size * kernel_size * c)
    return out


def correlation_torch(f1: torch.Tensor, f2: torch.Tensor, md: int = MD) -> torch.Tensor:
    """Fast path of :func:`correlation_np` for the PWC instantiation (k=1,s=1,pad=md)."""
    n, c, h, w = f1.shape
    p2 = F.pad(f2, (md, md, md, md))
    out = f1.new_empty((n, (2 * md + 1) ** 2, h, w))
    k = 0
    for dy in range(2 * md + 1):
        for dx in range(2 * md + 1):
            out[:, k] = (f1 * p2[:, :, dy:dy + h, dx:dx + w]).sum(1) / c
            k += 1
    return out


# --------------------------------------------------------------------------------------
def warp(x: torch.Tensor, flo: torch.Tensor) -> torch.Tensor:
    """Backward warp of ``x`` (image-2 features) by ``flo`` + validity mask, PWCNet.py:140-176."""
    b, c, h, w = x.shape
    xx = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w).expand(b, 1, h, w)
    yy = torch.arange(h, dtype=torch.float32).view(1, 1, h, 1).expand(b, 1, h, w)
    gx = 2.0 * (xx + flo[:, 0:1]) / max(w - 1, 1) - 1.0
    gy = 2.0 * (yy + flo[:, 1:2]) / max(h - 1, 1) - 1.0
    grid = torch.cat((gx, gy), 1).permute(0, 2, 3, 1)
    out = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    mask = F.grid_sample(torch.ones_like(x), grid, mode="bilinear", padding_mode="zeros",
                         align_corners=True)
    mask = (mask >= 0.9999).to(x.dtype)          # :173-174  (<0.9999 -> 0, rest -> 1)
    return out * mask


def _conv_lrelu(sd, name, x, stride=1, dilation=1):
    y = F.conv2d(x, sd[f"{name}.0.weight"], sd[f"{name}.0.bias"], stride=stride,
                 padding=dilation, dilation=dilation)
    return F.leaky_relu(y, 0.1)


def pyramid(sd, im: torch.Tensor) -> "dict[int, torch.Tensor]":
    """PWCNet.py:183-194 for one image."""
    strides = {name: s for name, _, _, s in PYRAMID}
    feats, x = {}, im
    for lvl in range(1, 7):
        for name in PYRAMID_ORDER[lvl]:
            x = _conv_lrelu(sd, name, x, stride=strides[name])
        feats[lvl] = x
    return feats


def pwc_forward(sd: "dict[str, torch.Tensor]", x: torch.Tensor, intermediates: bool = False):
    """``x`` [B,6,H,W] (H,W multiples of 64) -> flow2 [B,2,H/4,W/4]   (PWCNet.py:179-272, eval)."""
    c1 = pyramid(sd, x[:, :3])
    c2 = pyramid(sd, x[:, 3:])
    inter = {}
    up_flow = up_feat = None
    feat = None
    for lvl in (6, 5, 4, 3, 2):
        if lvl == 6:
            f2 = c2[lvl]
        else:
            f2 = warp(c2[lvl], up_flow * FLOW_SCALE[lvl])
            inter[f"warp{lvl}"] = f2
        corr = F.leaky_relu(correlation_torch(c1[lvl], f2), 0.1)
        inter[f"corr{lvl}"] = corr
        feat = corr if lvl == 6 else torch.cat((corr, c1[lvl], up_flow, up_feat), 1)
        for i in range(len(DENSE_GROWTH)):
            feat = torch.cat((_conv_lrelu(sd, f"conv{lvl}_{i}", feat), feat), 1)   # prepend
        flow = F.conv2d(feat, sd[f"predict_flow{lvl}.weight"], sd[f"predict_flow{lvl}.bias"],
                        padding=1)
        inter[f"flow{lvl}"] = flow
        if lvl != 2:
            up_flow = F.conv_transpose2d(flow, sd[f"deconv{lvl}.weight"], sd[f"deconv{lvl}.bias"],
                                         stride=2, padding=1)
            up_feat = F.conv_transpose2d(feat, sd[f"upfeat{lvl}.weight"], sd[f"upfeat{lvl}.bias"],
                                         stride=2, padding=1)
    y = feat
    for name, _, _, dil in CONTEXT:
        y = _conv_lrelu(sd, name, y, dilation=dil)
    flow2 = flow + F.conv2d(y, sd["dc_conv7.weight"], sd["dc_conv7.bias"], padding=1)
    if intermediates:
        inter.update({f"c1{l}": c1[l] for l in c1})
        inter.update({f"c2{l}": c2[l] for l in c2})
        return flow2, inter
    return flow2


def synth_frame_pair(h: int, w: int, seed: int = 1234, shift=(1.5, -0.75)) -> torch.Tensor:
    """Smooth seeded noise frame + sub-pixel translated copy -> [1,6,h,w] in [0,1]."""
    rng = np.random.default_rng(seed)
    lo = rng.random((3, h // 8 + 3, w // 8 + 3), dtype=np.float32)
    t = torch.from_numpy(lo)[None]
    big = F.interpolate(t, size=(h + 16, w + 16), mode="bicubic", align_corners=True).clamp(0, 1)
    im1 = big[:, :, 8:8 + h, 8:8 + w]
    ys = torch.arange(h, dtype=torch.float32) + 8 + shift[1]
    xs = torch.arange(w, dtype=torch.float32) + 8 + shift[0]
    gy = 2 * ys / (h + 15) - 1
    gx = 2 * xs / (w + 15) - 1
    grid = torch.stack(torch.meshgrid(gy, gx, indexing="ij")[::-1], -1)[None]
    im2 = F.grid_sample(big, grid, mode="bilinear", align_corners=True)
    return torch.cat((im1, im2), 1).contiguous()
