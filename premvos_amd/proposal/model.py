"""proposal_net forward on MI355X: class-agnostic ResNet-101-C4 Faster R-CNN (`train.py --forward`).

Reference graph: proposal_net/train.py:107-309 (Model._build_graph, inference branch) built from
basemodel.py:29-99 and model.py:17-51,113-217,300-395,438-491,551-565.

Design: one fixed launch list per resized image shape (captured into a HIP graph): NHWC activations,
frozen BatchNorm folded into the conv weights/bias, ReLU and the residual add fused into the conv
epilogue, RPN class+box 1x1 heads fused into one 75-channel conv (its NHWC output IS the
fHxfWxNA(x4) layout the reference transposes to), anchors/decode/top-k/clip/NMS in ONE kernel,
RoIAlign (crop_and_resize + avg-pool) in one kernel, the three FC heads as one 1x1 conv, the
inference tail (softmax/decode/clip/threshold/NMS/top-k) in one kernel.  Data-dependent counts
(<=100 RoIs, <=20 detections) stay on the device; buffers are fixed-size so the graph is static.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib, arena, ops
from ..ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, NHWC

RESNET_NUM_BLOCK = (3, 4, 23, 3)           # config.py:61
ANCHOR_STRIDE = 16
ANCHOR_SIZES = (32, 64, 128, 256, 512)
ANCHOR_RATIOS = (0.5, 1.0, 2.0)
NUM_ANCHOR = 15
MAX_SIZE = 1333
BBOX_DECODE_CLIP = float(np.float32(np.log(MAX_SIZE / 16.0)))   # config.py:77
TEST_PRE_NMS_TOPK, TEST_POST_NMS_TOPK = 1000, 100               # config.py:101,106
RPN_PROPOSAL_NMS_THRESH, RPN_MIN_SIZE = 0.7, 0.0
FASTRCNN_BBOX_REG_WEIGHTS = (10.0, 10.0, 5.0, 5.0)
FASTRCNN_NMS_THRESH, RESULT_SCORE_THRESH, RESULTS_PER_IM = 0.5, 0.5, 20
NUM_CLASS, SECOND_NUM_CLASS = 2, 81
BN_EPS = 1e-5


def cell_anchors() -> np.ndarray:
    """The NUM_RATIO x NUM_SCALE anchors of cell (0,0): utils/generate_anchors.py:40-100 with
    scales = sizes/stride, then x2,y2 += 1 (data.py:34-74) -> float32 [15,4]."""
    base = ANCHOR_STRIDE
    scales = np.array(ANCHOR_SIZES, np.float64) / base
    out = []
    w = h = float(base)
    xc = yc = 0.5 * (base - 1)
    for r in ANCHOR_RATIOS:
        ws = np.round(np.sqrt(w * h / r))
        hs = np.round(ws * r)
        for s in scales:
            sw, sh = ws * s, hs * s
            out.append([xc - 0.5 * (sw - 1), yc - 0.5 * (sh - 1), xc + 0.5 * (sw - 1), yc + 0.5 * (sh - 1)])
    a = np.array(out, np.float32)
    a[:, 2:] += 1
    return a


def _fold_bn(bn: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    scale = bn["gamma"].double() / torch.sqrt(bn["var"].double() + BN_EPS)
    bias = bn["beta"].double() - bn["mean"].double() * scale
    return scale.float(), bias.float()


class _Plan:
    def __init__(self, net: "ProposalNet", b: int, h: int, w: int):
        dev = net.device
        self.b, self.h, self.w = b, h, w
        # activation memory by liveness (premvos_amd/arena.py): the launch list is built twice -- shapes and lifetimes, then on
        # the packed arena (the bottleneck chain of a ResNet group rotates through the bytes of a handful of tensors)
        self.arena = arena.two_pass(dev, lambda A: self._build(net, b, h, w, A))
        self.ws_splitk = ops.assign_workspace(ops.autotune(self.tune_descs, dev) or self.tune_descs, dev)
        self.graph: Optional[torch.cuda.CUDAGraph] = None

    def _build(self, net: "ProposalNet", b: int, h: int, w: int, A: "arena.Arena"):
        dev = net.device
        P, lib = net.packed, _lib.load()
        steps: List = []
        self.flops: Dict[str, float] = {}
        self.descs: List = []

        def alloc(n, hh, ww, c):
            ps = (c + 3) // 4 * 4
            return NHWC(A.alloc(n, hh, ww, ps, "f32", pooled=ps == c), c=c)

        S8 = net.packed_s8
        self.tune_descs: List = []     # the launches premvos_conv2d_f32 runs (ops.autotune configures these; S8 convs have one kernel)

        def alloc_s8(n, hh, ww, c):
            assert c % 8 == 0
            return NHWC(A.alloc(n, hh, ww, c, "s8"), c=c, layout="s8")

        def release(*vs):              # the last launch that reads these tensors has been appended
            for v in vs:
                if v is not None:
                    A.release(v)

        def conv(x, name, out, uid=None, out_s8=None, res_s8=None, **kw):
            """``x`` in the resident split layout S8 (bf16x3 mode) -> csrc/conv_bf16x3_s8.hip (fp32 ``out`` and / or S8 ``out_s8``;
            ``res_s8``: the residual read from an S8 tensor); fp32 ``x`` -> premvos_conv2d_f32."""
            key = "conv:" + (uid or name)
            if x.layout == "s8":
                pk = S8[name]
                d = ops.conv_s8_desc(x, pk, out, out_s8, **kw)
                d.tile_hint = ops.S8_HINT
                o = out if out is not None else out_s8
                steps.append((key, lambda d=d, x=x, pk=pk, o8=out_s8, r8=res_s8: ops.run_s8(d, x, pk, o8, res_s8=r8)))
                self.split_layers += 1
            else:
                assert out_s8 is None and res_s8 is None
                pk = P[name]
                d = ops.conv_desc(x, pk, out, **kw)
                self.tune_descs.append(d)
                o = out
                steps.append((key, lambda d=d: ops.run_desc(d)))
            self.descs.append(d)
            self.flops[key] = 2.0 * o.n * o.h * o.w * pk.kh * pk.kw * pk.cin * pk.cout

        self.split_layers = 0          # convs on the S8 kernel (bf16x3 mode)
        self.img = alloc(b, h, w, 3)
        # conv0: pad [2,3] + 7x7 s2 VALID + BN + ReLU; pool0: pad [0,1] + 3x3 s2 VALID  (basemodel.py:79-82)
        h0, w0 = ops.out_size(h, 7, 2, 2, 3), ops.out_size(w, 7, 2, 2, 3)
        c0 = alloc(b, h0, w0, 64)
        conv(self.img, "conv0", c0, stride=(2, 2), pad=(2, 2), act=ACT_RELU)
        hp, wp = ops.out_size(h0, 3, 2, 0, 1), ops.out_size(w0, 3, 2, 0, 1)
        x = alloc(b, hp, wp, 64)

        def pool(i=c0, o=x):
            _lib.check(lib.premvos_maxpool_f32(i.ptr, i.ps, i.n, i.h, i.w, i.c, o.ptr, o.ps, o.h, o.w, 3, 2, 0, 0,
                                               0.0, _lib.current_stream()), "maxpool")
        steps.append(("maxpool", pool))
        release(c0)

        def group(x: Optional[NHWC], g: int, feat: int, count: int, stride: int, tag: str = "", x8: Optional[NHWC] = None,
                  last_s8: bool = False, last_f32: bool = True):
            """One ResNet group (basemodel.py:62-72).  fp32 mode: every tensor is floats.  bf16x3 mode, groups >= net.s8_from: the
            bottleneck chain lives in S8 ONLY -- conv1 / conv2 / conv3 read and write S8, the residual of an identity block is
            read from the previous block's S8 output (hi + lo), a convshortcut's output is fp32 -- and only the last block of a
            group whose output a float kernel reads (RoIAlign / RPN after group 2, the average pool after conv5: ``last_f32``)
            also writes floats.  ``x8``: the S8 form of the group's input when the producer wrote one; else one split pass over
            ``x``.  Returns (y or None, y8 or None)."""
            s8 = net.s8 and g >= net.s8_from
            shape = (x if x is not None else x8)
            n_, h_, w_, c_ = shape.n, shape.h, shape.w, shape.c
            own_x = own_x8 = False     # the group's own input belongs to the caller
            for i in range(count):
                p = f"group{g}/block{i}"
                s = stride if i == 0 else 1
                last = i + 1 == count
                if s8 and x8 is None:                   # entry of the chain: one split pass over the fp32 tensor
                    x8 = alloc_s8(n_, h_, w_, c_)
                    own_x8 = True
                    steps.append((f"split8:{tag}{p}", lambda i_=x, o_=x8: ops.split8(i_, o_)))
                xin = x8 if s8 else x
                mk = alloc_s8 if s8 else alloc
                t1 = mk(n_, h_, w_, feat)
                o1 = dict(out=None, out_s8=t1) if s8 else dict(out=t1)
                conv(xin, p + "/conv1", uid=tag + p + "/conv1", act=ACT_RELU, **o1)
                if s == 2:        # pad [0,1] + VALID stride 2  (basemodel.py:54-56)
                    ho, wo = ops.out_size(h_, 3, 2, 0, 1), ops.out_size(w_, 3, 2, 0, 1)
                    t2 = mk(n_, ho, wo, feat)
                    o2 = dict(out=None, out_s8=t2) if s8 else dict(out=t2)
                    conv(t1, p + "/conv2", uid=tag + p + "/conv2", stride=(2, 2), pad=(0, 0), act=ACT_RELU, **o2)
                else:
                    ho, wo = h_, w_
                    t2 = mk(n_, h_, w_, feat)
                    o2 = dict(out=None, out_s8=t2) if s8 else dict(out=t2)
                    conv(t1, p + "/conv2", uid=tag + p + "/conv2", pad=(1, 1), act=ACT_RELU, **o2)
                res, res8 = x, None
                if p + "/convshortcut" in P or p + "/convshortcut" in S8:   # 1x1 stride s on x[:, :, :-1, :-1] == reading pixel (s*oy, s*ox)
                    res = alloc(n_, ho, wo, feat * 4)
                    conv(xin, p + "/convshortcut", res, uid=tag + p + "/convshortcut", stride=(s, s))
                elif s8:
                    res, res8 = None, x8
                want_f32 = not s8 or (last and last_f32)
                y = alloc(n_, ho, wo, feat * 4) if want_f32 else None
                y8 = alloc_s8(n_, ho, wo, feat * 4) if s8 and (not last or last_s8) else None
                conv(t2, p + "/conv3", y, uid=tag + p + "/conv3", out_s8=y8, res=res, res_s8=res8, act=ACT_RELU)     # relu(bn(conv3) + shortcut)
                # the block's input, its two inner tensors and a convshortcut's output are dead now
                release(t1, t2, res if res is not x else None, x if own_x else None, x8 if own_x8 else None)
                x, x8, h_, w_, c_, own_x, own_x8 = y, y8, ho, wo, feat * 4, True, True
            return x, x8

        nb = net.num_blocks
        s8g = [net.s8 and g >= net.s8_from for g in range(4)]
        x1, x18 = group(x, 0, 64, nb[0], 1, last_s8=s8g[1], last_f32=not s8g[1])
        release(x)
        x2, x28 = group(x1, 1, 128, nb[1], 2, x8=x18, last_s8=s8g[2], last_f32=not s8g[2])
        release(x1, x18)
        fm, fm8 = group(x2, 2, 256, nb[2], 2, x8=x28, last_s8=net.s8 and net.s8_rpn)
        release(x2, x28)
        self.featuremap = fm
        fh, fw = fm.h, fm.w
        # rpn_head (model.py:30-51): 3x3 + ReLU, then class(15) + box(60) as one 1x1 conv
        hid = alloc(b, fh, fw, 1024)
        conv(fm8 if fm8 is not None else fm, "rpn/conv0", hid, pad=(1, 1), act=ACT_RELU)
        release(fm8)
        self.rpn_out = alloc(b, fh, fw, 5 * NUM_ANCHOR)
        conv(hid, "rpn/heads", self.rpn_out)
        release(hid)
        R = TEST_POST_NMS_TOPK
        self.rois = A.side((b, R, 4), torch.float32)
        self.roi_scores = A.side((b, R), torch.float32)
        self.roi_idx = A.side((b, R), torch.int32)
        self.roi_count = A.side((b,), torch.int32)
        ca = net.cell_anchors_dev

        def rpn(o=self.rpn_out):
            _lib.check(lib.premvos_rpn_proposals_f32(
                o.ptr, o.ps, b, fh, fw, NUM_ANCHOR, 0, NUM_ANCHOR, ca.data_ptr(), float(ANCHOR_STRIDE), float(h),
                float(w), TEST_PRE_NMS_TOPK, R, RPN_PROPOSAL_NMS_THRESH, RPN_MIN_SIZE, BBOX_DECODE_CLIP,
                self.rois.data_ptr(), self.roi_scores.data_ptr(), self.roi_idx.data_ptr(), self.roi_count.data_ptr(),
                _lib.current_stream()), "rpn_proposals")
        steps.append(("rpn_proposals", rpn))
        roi = alloc(b * R, 14, 14, 1024)

        def ralign(o=roi):
            _lib.check(lib.premvos_roi_align_f32(fm.ptr, fm.ps, b, fh, fw, 1024, self.rois.data_ptr(),
                                                 self.roi_count.data_ptr(), R, 1.0 / ANCHOR_STRIDE, 14, o.ptr, o.ps,
                                                 _lib.current_stream()), "roi_align")
        steps.append(("roi_align", ralign))
        f5, _ = group(roi, 3, 512, nb[3], 2)         # resnet_conv5 (basemodel.py:92-99)
        release(roi)
        self.feat5 = f5
        gp = alloc(b * R, 1, 1, 2048)

        def gap(i=f5, o=gp):
            _lib.check(lib.premvos_global_avgpool_f32(i.ptr, i.ps, i.n, i.h * i.w, i.c, o.ptr, o.ps,
                                                      _lib.current_stream()), "gap")
        steps.append(("global_avgpool", gap))
        # (f5, gp, the head logits and the feature map stay to the end of the list: drivers and tests read them after a run)
        nh = NUM_CLASS + 4 * (NUM_CLASS - 1) + SECOND_NUM_CLASS
        self.head = alloc(b * R, 1, 1, nh)
        conv(gp, "heads", self.head)
        M = RESULTS_PER_IM
        self.final_boxes = A.side((b, M, 4), torch.float32)
        self.final_probs = A.side((b, M), torch.float32)
        self.final_idx = A.side((b, M), torch.int32)
        self.final_count = A.side((b,), torch.int32)

        def tail(hd=self.head):
            _lib.check(lib.premvos_frcnn_tail_f32(
                hd.ptr, hd.ps, self.rois.data_ptr(), self.roi_count.data_ptr(), b, R, float(h), float(w),
                RESULT_SCORE_THRESH, FASTRCNN_NMS_THRESH, M, BBOX_DECODE_CLIP, *FASTRCNN_BBOX_REG_WEIGHTS,
                self.final_boxes.data_ptr(), self.final_probs.data_ptr(), self.final_idx.data_ptr(),
                self.final_count.data_ptr(), _lib.current_stream()), "frcnn_tail")
        steps.append(("frcnn_tail", tail))

        # mask head (train.py:297-309, model.py:494-509) -- OFF in the shipped --forward pipeline (train.py:636-637)
        self.final_masks: Optional[NHWC] = None
        if net.mode_mask:
            mroi = alloc(b * M, 14, 14, 1024)

            def malign(o=mroi):
                _lib.check(lib.premvos_roi_align_f32(fm.ptr, fm.ps, b, fh, fw, 1024, self.final_boxes.data_ptr(),
                                                     self.final_count.data_ptr(), M, 1.0 / ANCHOR_STRIDE, 14, o.ptr,
                                                     o.ps, _lib.current_stream()), "roi_align(mask)")
            steps.append(("roi_align_mask", malign))
            mf5, _ = group(mroi, 3, 512, nb[3], 2, tag="mask:")   # the SAME conv5 weights (auto_reuse_variable_scope)
            release(mroi)
            up = alloc(b * M, 14, 14, 256)
            conv(mf5, "maskrcnn/deconv", up, act=ACT_RELU)
            release(mf5)
            self.final_masks = alloc(b * M, 14, 14, NUM_CLASS - 1)
            conv(up, "maskrcnn/conv", self.final_masks, act=ACT_SIGMOID)
            release(up)
        self.steps = steps

    def run(self, steps=None):
        for _, fn in (self.steps if steps is None else steps):
            fn()

    def capture(self, steps=None):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.run(steps)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):      # other host threads (IO lanes) keep using the GPU
            self.run(steps)
        if steps is None:
            self.graph = g
        return g


class ProposalNet:
    """Weights + per-shape plans.  ``weights`` uses the reference variable names with conv
    kernels in OIHW: 'conv0/W', 'group{g}/block{i}/conv{1,2,3}/W', '.../convshortcut/W', '<conv>/bn' = dict(gamma,
    beta, mean, var), 'rpn/{conv0,class,box}/{W,b}', 'fastrcnn/{class,box}/{W,b}', 'secondclassification/class/{W,b}'."""

    def __init__(self, weights: Dict[str, object], num_blocks: Sequence[int] = RESNET_NUM_BLOCK,
                 device=None, use_graph: bool = True, precision: Optional[str] = None,
                 mode_mask: bool = False):
        _lib.require_gpu()
        self.mode_mask = mode_mask
        self.precision = prec = precision or ops.default_precision()
        self.device, self.use_graph, self.num_blocks = _lib.resolve_device(device), use_graph, tuple(num_blocks)
        device = self.device
        self.packed: Dict[str, ops.PackedConv] = {}
        self._plans: Dict[tuple, _Plan] = {}
        self.cell_anchors_dev = torch.from_numpy(cell_anchors()).to(device)
        w = weights
        # bf16x3 (split-fp32) mode, round 4: from group ``s8_from`` on (PREMVOS_BF16X3_S8_FROM, default 1: group0's K = 64 layers are
        # HBM-bound on any pipe) every conv of the bottleneck chains, the RPN 3x3 and conv5 runs on csrc/conv_bf16x3_s8.hip with its
        # input resident in the split layout S8; conv0 and group0 stay on the fp32 kernels the shipped table tunes (packed with
        # precision="fp32" below), the 75- / 87-channel heads run on the on-the-fly bf16x3 kernel
        import os
        self.s8 = prec == "bf16x3" and os.environ.get("PREMVOS_BF16X3_SPLIT", "1") != "0"
        self.s8_from = int(os.environ.get("PREMVOS_BF16X3_S8_FROM", "1"))
        self.packed_s8: Dict[str, ops.PackedConvS8] = {}
        for name in [k[:-2] for k in w if k.endswith("/W") and (k[:-2] + "/bn") in w]:
            scale, bias = _fold_bn(w[name + "/bn"])
            if self.s8 and name.startswith("group") and int(name[5]) >= self.s8_from:
                self.packed_s8[name] = ops.pack_conv_s8(w[name + "/W"], bias, device, scale=scale)
            else:       # (with S8 chains: conv0 / the HBM-bound group-0 layers stay on the fp32 kernels the shipped table tunes)
                self.packed[name] = ops.pack_conv(w[name + "/W"], bias, device, scale=scale, precision="fp32" if self.s8 else prec)
        # the RPN 3x3 (1024 -> 1024, K = 9216) stays on fp32 Winograd F(4x4,3x3) by default: 4x fewer multiplies = 456 ... 467
        # TFLOP/s-equivalent against ~400 for three bf16 MFMAs per product on this shape (profiles/r04_s8_bench.txt); PREMVOS_S8_RPN=1
        self.s8_rpn = os.environ.get("PREMVOS_S8_RPN", "0") == "1"
        if self.s8 and self.s8_rpn:
            self.packed_s8["rpn/conv0"] = ops.pack_conv_s8(w["rpn/conv0/W"], w["rpn/conv0/b"], device)
        else:
            self.packed["rpn/conv0"] = ops.pack_conv(w["rpn/conv0/W"], w["rpn/conv0/b"], device, precision="fp32" if self.s8 else prec)
        self.packed["rpn/heads"] = ops.pack_conv(torch.cat([w["rpn/class/W"], w["rpn/box/W"]], 0),
                                                 torch.cat([w["rpn/class/b"], w["rpn/box/b"]], 0), device, precision=prec)
        hw = torch.cat([w["fastrcnn/class/W"], w["fastrcnn/box/W"], w["secondclassification/class/W"]], 0)
        hb = torch.cat([w["fastrcnn/class/b"], w["fastrcnn/box/b"], w["secondclassification/class/b"]], 0)
        self.packed["heads"] = ops.pack_conv(hw.view(hw.shape[0], hw.shape[1], 1, 1), hb, device, precision=prec)
        if mode_mask:
            self.packed["maskrcnn/deconv"] = ops.pack_deconv2x2s2(w["maskrcnn/deconv/W"], w["maskrcnn/deconv/b"], device,
                                                                  precision=prec)
            self.packed["maskrcnn/conv"] = ops.pack_conv(w["maskrcnn/conv/W"], w["maskrcnn/conv/b"], device, precision=prec)

    def plan(self, b: int, h: int, w: int) -> _Plan:
        key = (b, h, w)
        if key not in self._plans:
            if h // ANCHOR_STRIDE < 2 or w // ANCHOR_STRIDE < 2:
                raise ValueError("image too small for the stride-16 feature map")
            p = _Plan(self, b, h, w)
            if self.use_graph:
                p.capture()
            self._plans[key] = p
        return self._plans[key]

    def run_resized(self, img_bgr: torch.Tensor) -> _Plan:
        """img_bgr: uint8 [B,h,w,3] on the device, ALREADY resized (what the reference feeds the TF graph)."""
        b, h, w, _ = img_bgr.shape
        p = self.plan(b, h, w)
        _lib.check(_lib.load().premvos_proposal_preprocess_u8(img_bgr.contiguous().data_ptr(), b, h, w, p.img.ptr, h,
                                                              w, 0, _lib.current_stream()), "proposal_preprocess")
        if p.graph is not None:
            p.graph.replay()
        else:
            p.run()
        return p

    def outputs(self, p: _Plan, i: int = 0):
        """Host tuple in the order of get_model_output_names() (train.py:52-62) for image i of the batch."""
        n = int(p.final_count[i].item())
        boxes = p.final_boxes[i, :n].cpu().numpy()
        probs = p.final_probs[i, :n].cpu().numpy()
        idx = p.final_idx[i, :n].cpu().numpy().astype(np.int64)
        labels = np.ones((n,), np.int64)
        head = p.head.buf.view(p.b, TEST_POST_NMS_TOPK, -1)[i].cpu().numpy()
        cls = head[:, :2]
        e = np.exp(cls - cls.max(1, keepdims=True))
        label_probs = e / e.sum(1, keepdims=True)
        # train.py:287-288 quirk: gathered by *category* id (always 0 here), kept for interface parity
        final_posterior = label_probs[np.zeros((n,), np.int64)] if n else np.zeros((0, 2), np.float32)
        sec = head[:, 6:6 + SECOND_NUM_CLASS]
        es = np.exp(sec - sec.max(1, keepdims=True))
        sec_probs = es / es.sum(1, keepdims=True)
        second_final_posterior = sec_probs[np.zeros((n,), np.int64)] if n else np.zeros((0, SECOND_NUM_CLASS), np.float32)
        second_final_labels = (final_posterior.argmax(-1) + 1) if n else np.zeros((0,), np.int64)
        return boxes, probs, labels, final_posterior, second_final_labels, second_final_posterior, idx

    def masks(self, p: _Plan, i: int = 0) -> np.ndarray:
        """'final_masks' of get_model_output_names() when MODE_MASK: [n,14,14] sigmoid probabilities."""
        n = int(p.final_count[i].item())
        m = p.final_masks.buf.view(p.b, RESULTS_PER_IM, 14, 14, -1)[i, :n, :, :, 0]
        return m.cpu().numpy()
