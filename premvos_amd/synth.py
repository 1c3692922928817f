"""Synthetic stand-ins for what the build container does not have: the released checkpoints (simple_run.sh:12-17
downloads 3 GB at run time) and DAVIS frames.  Used by bench.py, tools/ and smoke runs; built from the product's own
network tables, so nothing here touches ``oracle/`` (tests keep using the oracle's generators for parity work).

Weights follow SURVEY.md 8(d): He-normal kernels, BatchNorm gamma around 1 (0.25 on the last conv of a residual branch so
that 30-100 stacked units stay O(1)), beta and mean ~ N(0, 0.1), variance ~ U(0.5, 1.5); one RNG stream per tensor name,
so a tensor does not depend on construction order.  Names and layouts are the ones the nets' ``load`` paths expect
(conv kernels OIHW).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def _gen(name: str, seed: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) + 1000003 * seed) & 0x7FFFFFFF)


def _kernel(name: str, cout: int, cin: int, k: int, seed: int, gain: float = 2.0) -> torch.Tensor:
    return torch.randn((cout, cin, k, k), generator=_gen(name, seed)) * math.sqrt(gain / (cin * k * k))


def _vec(name: str, n: int, seed: int, std: float) -> torch.Tensor:
    return torch.randn(n, generator=_gen(name, seed)) * std


def _bn(name: str, c: int, seed: int, gamma=(0.8, 1.2)) -> Dict[str, torch.Tensor]:
    g = _gen(name + "#bn", seed)
    return {"gamma": torch.rand(c, generator=g) * (gamma[1] - gamma[0]) + gamma[0],
            "beta": torch.randn(c, generator=g) * 0.1, "mean": torch.randn(c, generator=g) * 0.1,
            "var": torch.rand(c, generator=g) + 0.5}


# ---------------------------------------------------------------------------------------------------------------------
def pwc_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """The reference module's state_dict (PWCNet.py:50-131): ``<name>.0.weight/bias`` for conv()+LeakyReLU pairs, bare
    ``predict_flow*``, ``deconv*``, ``upfeat*`` ([in, out, 4, 4]), ``dc_conv7``."""
    from .flow.pwcnet import CONTEXT, FEAT, GROWTH, GROW_SUM, PYR, _od
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, suffix=".0"):
        sd[f"{name}{suffix}.weight"] = _kernel(name, cout, cin, 3, seed)
        sd[f"{name}{suffix}.bias"] = _vec(name + "#b", cout, seed, 0.05)

    cin = 3
    for lvl in range(1, 7):
        for name in PYR[lvl]:
            conv(name, FEAT[lvl], cin)
            cin = FEAT[lvl]
    for lvl in (6, 5, 4, 3, 2):
        c = _od(lvl)
        for i, g in enumerate(GROWTH):
            conv(f"conv{lvl}_{i}", g, c)
            c += g
        conv(f"predict_flow{lvl}", 2, c, suffix="")
        # transposed convs: only 4 of the 16 taps reach one output pixel
        sd[f"deconv{lvl}.weight"] = torch.randn((2, 2, 4, 4), generator=_gen(f"deconv{lvl}", seed)) * math.sqrt(2.0 / (2 * 4))
        sd[f"deconv{lvl}.bias"] = _vec(f"deconv{lvl}#b", 2, seed, 0.05)
        if lvl != 2:
            sd[f"upfeat{lvl}.weight"] = torch.randn((c, 2, 4, 4), generator=_gen(f"upfeat{lvl}", seed)) * math.sqrt(2.0 / (c * 4))
            sd[f"upfeat{lvl}.bias"] = _vec(f"upfeat{lvl}#b", 2, seed, 0.05)
    cin = _od(2) + GROW_SUM
    for name, cout, _ in CONTEXT:
        conv(name, cout, cin)
        cin = cout
    conv("dc_conv7", 2, 32, suffix="")
    return sd


def proposal_weights(seed: int = 0, num_blocks: Sequence[int] = None) -> Dict[str, object]:
    """tensorpack scopes (basemodel.py:29-99, model.py:30-51,377-395,494-509,551-565): ``<scope>/W`` + ``<scope>/bn`` dicts
    for the backbone, ``W``/``b`` for RPN and the heads (FC matrices [out, in])."""
    from .proposal.model import NUM_ANCHOR, NUM_CLASS, RESNET_NUM_BLOCK, SECOND_NUM_CLASS
    nb = RESNET_NUM_BLOCK if num_blocks is None else num_blocks
    w: Dict[str, object] = {"conv0/W": _kernel("conv0", 64, 3, 7, seed), "conv0/bn": _bn("conv0", 64, seed)}
    cin = 64
    for g, (feat, cnt) in enumerate(zip((64, 128, 256, 512), nb)):
        for i in range(cnt):
            p = f"group{g}/block{i}"
            for name, co, ci, k, gain, gam in (("conv1", feat, cin, 1, 2.0, (0.8, 1.2)), ("conv2", feat, feat, 3, 2.0, (0.8, 1.2)),
                                               ("conv3", 4 * feat, feat, 1, 1.0, (0.15, 0.35))):
                w[f"{p}/{name}/W"] = _kernel(f"{p}/{name}", co, ci, k, seed, gain)
                w[f"{p}/{name}/bn"] = _bn(f"{p}/{name}", co, seed, gam)
            if cin != 4 * feat:
                w[f"{p}/convshortcut/W"] = _kernel(f"{p}/convshortcut", 4 * feat, cin, 1, seed, 1.0)
                w[f"{p}/convshortcut/bn"] = _bn(f"{p}/convshortcut", 4 * feat, seed)
            cin = 4 * feat
    w["rpn/conv0/W"], w["rpn/conv0/b"] = _kernel("rpn/conv0", 1024, 1024, 3, seed), _vec("rpn/conv0#b", 1024, seed, 0.05)
    w["rpn/class/W"], w["rpn/class/b"] = _kernel("rpn/class", NUM_ANCHOR, 1024, 1, seed, 8.0), _vec("rpn/class#b", NUM_ANCHOR, seed, 0.5)
    w["rpn/box/W"], w["rpn/box/b"] = _kernel("rpn/box", 4 * NUM_ANCHOR, 1024, 1, seed, 0.3), _vec("rpn/box#b", 4 * NUM_ANCHOR, seed, 0.1)
    cw = torch.randn((NUM_CLASS, 2048), generator=_gen("fastrcnn/class", seed)) * 0.4
    w["fastrcnn/class/W"], w["fastrcnn/class/b"] = cw - cw.mean(dim=1, keepdim=True), _vec("fastrcnn/class#b", NUM_CLASS, seed, 0.1)
    w["fastrcnn/box/W"] = torch.randn((4 * (NUM_CLASS - 1), 2048), generator=_gen("fastrcnn/box", seed)) * 0.05
    w["fastrcnn/box/b"] = _vec("fastrcnn/box#b", 4 * (NUM_CLASS - 1), seed, 0.1)
    w["maskrcnn/deconv/W"] = torch.randn((2048, 256, 2, 2), generator=_gen("maskrcnn/deconv", seed)) * math.sqrt(2.0 / 2048)
    w["maskrcnn/deconv/b"] = _vec("maskrcnn/deconv#b", 256, seed, 0.05)
    w["maskrcnn/conv/W"], w["maskrcnn/conv/b"] = _kernel("maskrcnn/conv", NUM_CLASS - 1, 256, 1, seed, 8.0), _vec("maskrcnn/conv#b", NUM_CLASS - 1, seed, 0.3)
    w["secondclassification/class/W"] = torch.randn((SECOND_NUM_CLASS, 2048), generator=_gen("second", seed)) * 0.05
    w["secondclassification/class/b"] = _vec("second#b", SECOND_NUM_CLASS, seed, 0.1)
    return w


def refinement_weights(seed: int = 0, num_middle: int = 16) -> Dict[str, object]:
    """slim scopes of the DeepLabv3+ graph without the ``xception_65/`` prefix (network/deeplab/model.py:61-66,
    core/xception.py:172-177,275,282): ``<scope>/weights``, ``<scope>/depthwise_weights`` [C,1,3,3], ``<scope>/BatchNorm``."""
    from .refinement.model import module_plan
    w: Dict[str, object] = {}

    def conv(scope, cout, cin, k=1, gain=2.0, gamma=(0.8, 1.2)):
        w[scope + "/weights"] = _kernel(scope, cout, cin, k, seed, gain)
        w[scope + "/BatchNorm"] = _bn(scope, cout, seed, gamma)

    def depthwise(scope, c, gain=2.0):
        w[scope + "/depthwise_weights"] = torch.randn((c, 1, 3, 3), generator=_gen(scope, seed)) * math.sqrt(gain / 9.0)
        w[scope + "/BatchNorm"] = _bn(scope, c, seed)

    conv("entry_flow/conv1_1", 32, 4, 3)
    conv("entry_flow/conv1_2", 64, 32, 3)
    for prefix, cin, depths, skip, relu_in, _, _ in module_plan(num_middle):
        c = cin
        for i, d in enumerate(depths):
            s = f"{prefix}/separable_conv{i + 1}"
            depthwise(s + "_depthwise", c, 2.0 if relu_in else 1.0)
            closes_branch = i == 2 and skip != "none"
            conv(s + "_pointwise", d, c, 1, 2.0 if (relu_in or i < 2) else 1.0, (0.15, 0.35) if closes_branch else (0.8, 1.2))
            c = d
        if skip == "conv":
            conv(prefix + "/shortcut", depths[-1], cin, 1, 1.0)
    conv("image_pooling", 256, 2048)
    conv("aspp0", 256, 2048)
    for i in (1, 2, 3):
        depthwise(f"aspp{i}_depthwise", 2048)
        conv(f"aspp{i}_pointwise", 256, 2048)
    conv("concat_projection", 256, 1280)
    conv("decoder/feature_projection0", 48, 256)
    for j, cin in ((0, 304), (1, 256)):
        depthwise(f"decoder/decoder_conv{j}_depthwise", cin)
        conv(f"decoder/decoder_conv{j}_pointwise", 256, cin)
    w["logits/features/weights"] = _kernel("logits/features", 2, 256, 1, seed, 1.0)
    w["logits/features/biases"] = _vec("logits/features#b", 2, seed, 0.1)
    return w


# ---------------------------------------------------------------------------------------------------------------------
def frame_pair(h: int, w: int, seed: int = 1234, shift=(1.5, -0.75)) -> torch.Tensor:
    """Smooth seeded noise frame + its sub-pixel translated successor -> float [1,6,h,w] in [0,1]: low-pass
    noise on an 8x coarser lattice, bicubic up-sampling, second frame = bilinear resampling at (x + dx, y + dy)."""
    rng = np.random.default_rng(seed)
    coarse = torch.from_numpy(rng.random((3, h // 8 + 3, w // 8 + 3), dtype=np.float32))[None]
    big = F.interpolate(coarse, size=(h + 16, w + 16), mode="bicubic", align_corners=True).clamp(0, 1)
    gy = 2 * (torch.arange(h, dtype=torch.float32) + 8 + shift[1]) / (h + 15) - 1
    gx = 2 * (torch.arange(w, dtype=torch.float32) + 8 + shift[0]) / (w + 15) - 1
    grid = torch.stack(torch.meshgrid(gy, gx, indexing="ij")[::-1], -1)[None]
    second = F.grid_sample(big, grid, mode="bilinear", align_corners=True)
    return torch.cat((big[:, :, 8:8 + h, 8:8 + w], second), 1).contiguous()


def video_frames(batch: int, h: int, w: int, rank: int = 0):
    """``batch`` (frame t, frame t+1) pairs as uint8 RGB [B,H,W,3] each; different frames per rank."""
    first, second = [], []
    for i in range(batch):
        pair = frame_pair(h, w + (-w) % 2, seed=1234 + 100 * rank + i, shift=(1.5 + 0.25 * i, -0.75))
        fr = (pair[0, :, :, :w].permute(1, 2, 0) * 255).round().to(torch.uint8)
        first.append(fr[..., :3])
        second.append(fr[..., 3:])
    return torch.stack(first).contiguous(), torch.stack(second).contiguous()


def clip_frames(first: int, end: int, h: int, w: int, seed: int = 77) -> torch.Tensor:
    """Frames [first, end) of ONE seeded synthetic video as uint8 RGB [n,H,W,3]: a smooth noise field drifting by a sub-pixel
    step per frame, so frame t is the same bytes whichever rank (and whichever range) asks for it -- what the strong-scaling
    bench shards (bench.py --scaling strong)."""
    rng = np.random.default_rng(seed)
    pad = 40
    coarse = torch.from_numpy(rng.random((3, h // 8 + 3 + pad // 4, w // 8 + 3 + pad // 4), dtype=np.float32))[None]
    big = F.interpolate(coarse, size=(h + 2 * pad, w + 2 * pad), mode="bicubic", align_corners=True).clamp(0, 1)
    out = []
    for t in range(first, end):
        dx, dy = pad * math.sin(0.011 * t + 0.3) * 0.9, pad * math.cos(0.007 * t) * 0.9     # stays inside the padded field
        gy = 2 * (torch.arange(h, dtype=torch.float32) + pad + dy) / (h + 2 * pad - 1) - 1
        gx = 2 * (torch.arange(w, dtype=torch.float32) + pad + dx) / (w + 2 * pad - 1) - 1
        grid = torch.stack(torch.meshgrid(gy, gx, indexing="ij")[::-1], -1)[None]
        fr = F.grid_sample(big, grid, mode="bilinear", align_corners=True)[0]
        out.append((fr.permute(1, 2, 0) * 255).round().to(torch.uint8))
    return torch.stack(out).contiguous() if out else torch.zeros((0, h, w, 3), dtype=torch.uint8)


def clip_boxes(first: int, end: int, per_frame: int, h: int, w: int, seed: int = 4321) -> torch.Tensor:
    """[n,P,4] boxes of frames [first, end) of the clip, a function of the frame index only."""
    return torch.cat([boxes(1, per_frame, h, w, rank=seed + 7919 * t - 4321) for t in range(first, end)]) if end > first \
        else torch.zeros((0, per_frame, 4), dtype=torch.float32)


def boxes(batch: int, per_frame: int, h: int, w: int, rank: int = 0) -> torch.Tensor:
    """[B,P,4] (y0,x0,y1,x1): seeded uniform boxes with w,h in [40,400] clipped to the frame."""
    rng = np.random.default_rng(4321 + rank)
    wh = np.minimum(rng.uniform(40, 400, (batch, per_frame, 2)), [w, h])
    xy = rng.uniform(0, 1, (batch, per_frame, 2)) * (np.array([w, h]) - wh)
    return torch.tensor(np.stack([xy[..., 1], xy[..., 0], xy[..., 1] + wh[..., 1], xy[..., 0] + wh[..., 0]], -1),
                        dtype=torch.float32)
