"""refinement_net forward on MI355X: DeepLabv3+ (Xception-65, OS16, ASPP 6/12/18, decoder OS4) on 385x385 crops.

Reference graph: refinement_net/network/deeplab/{DeepLabV3Plus.py:6-39, model.py:200-707, core/xception.py:70-560,
core/feature_extractor.py:90-116,202} + input pipeline datasets/{Dataset.py:48-56,141-186, Resize.py:150-193} +
output layer network/SegmentationOutputLayers.py:35-61,106-135.

Design: the reference feeds ONE box per session.run (batch 1, re-uploading the whole frame per box and
pulling two full-frame maps back); here all boxes of a frame form one batch: one uploaded uint8 frame, crops
cut on the GPU, one fixed launch list (HIP graph) over [P,385,385,4], masks and conf_scores left in HBM.
BatchNorm is folded into the depthwise / pointwise weights, the module's leading ReLU is applied on the
depthwise kernel's loads, the residual add rides the third pointwise conv's epilogue, ASPP branches and the
decoder inputs are written straight into their concat buffers.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch

from .. import _lib, arena, ops
from ..ops import ACT_NONE, ACT_RELU, NHWC

INPUT_SIZE = 385
EPS_BACKBONE, EPS_HEAD = 1e-3, 1e-5
ATROUS_RATES = (6, 12, 18)
BLOCKS = (   # scope, depth_list, skip, relu_inside, units, stride      (core/xception.py:506-551)
    ("entry_flow/block1", (128, 128, 128), "conv", False, 1, 2),
    ("entry_flow/block2", (256, 256, 256), "conv", False, 1, 2),
    ("entry_flow/block3", (728, 728, 728), "conv", False, 1, 2),
    ("middle_flow/block1", (728, 728, 728), "sum", False, 16, 1),
    ("exit_flow/block1", (728, 1024, 1024), "conv", False, 1, 2),
    ("exit_flow/block2", (1536, 1536, 2048), "none", True, 1, 1),
)
DECODER_SKIP = "entry_flow/block2/unit_1/xception_module/separable_conv2"


def module_plan(num_middle: int = 16):
    """(prefix, cin, depths, skip, relu_inside, stride, rate) per module with the stride->atrous switch of
    stack_blocks_dense (core/xception.py:330-345) for output_stride 16."""
    mods, cin, cur, rate = [], 64, 1, 1
    for scope, depths, skip, relu_in, units, stride in BLOCKS:
        for u in range(num_middle if scope.startswith("middle") else units):
            name = f"{scope}/unit_{u + 1}/xception_module"
            if cur == 8:
                mods.append((name, cin, depths, skip, relu_in, 1, rate))
                rate *= stride
            else:
                mods.append((name, cin, depths, skip, relu_in, stride, 1))
                cur *= stride
            cin = depths[-1]
    return mods


def _fold(bn: Dict[str, torch.Tensor], eps: float):
    scale = bn["gamma"].double() / torch.sqrt(bn["var"].double() + eps)
    return scale.float(), (bn["beta"].double() - bn["mean"].double() * scale).float()


class PackedDW:
    def __init__(self, w: torch.Tensor, bn: Dict[str, torch.Tensor], eps: float, device: str):
        c = w.shape[0]
        self.c, self.c_pad = c, (c + 3) // 4 * 4
        scale, bias = _fold(bn, eps)
        wk = (w.float().view(c, 9) * scale.view(c, 1)).t().contiguous()      # [9][c]
        full = torch.zeros((9, self.c_pad), dtype=torch.float32)
        full[:, :c] = wk
        b = torch.zeros(self.c_pad, dtype=torch.float32)
        b[:c] = bias
        self.wgt, self.bias = full.to(device), b.to(device)


class _Plan:
    def __init__(self, net: "RefinementNet", P: int, H: int, W: int, with_posterior: bool, frames: int = 1,
                 packed: bool = False, lane: int = 0):
        """One launch list for ``frames`` frames x ``P`` boxes each: the crops of all frames form ONE batch of the network
        (bigger GEMM M -> fewer partly filled waves of tiles); crop extraction and un-cropping run per frame.
        ``packed``: ``P`` is the TOTAL number of box slots of up to ``frames`` frames; which slots belong to which frame is
        set per call (``layout``: the boxes of a frame are a contiguous run), so a group of frames with 20, 23, 25 and 22
        proposals runs 96 crops, not 4 x 26 (eager launches only: the slot offsets are launch arguments)."""
        dev = net.device
        self.P, self.H, self.W, self.G, self.packed = P, H, W, frames, packed
        # activation memory by liveness (premvos_amd/arena.py): the launch list is built twice -- shapes and lifetimes, then on
        # the packed arena (up to round 4 same-SHAPE buffers were pooled: the 193 x 193 tensors of the entry flow, 15 GB of a
        # 160-crop plan, sat idle for the rest of the list)
        # ... and the plans of one lane (one per packed slot count / group of frames; a lane runs one plan at a time, stream-ordered)
        # share the lane's bytes
        self.arena = arena.two_pass(dev, lambda A: self._build(net, P, H, W, with_posterior, frames, packed, A),
                                    shared=net._lane_bytes.setdefault(lane, {}) if arena.enabled() else None)
        self.ws_splitk = ops.assign_workspace(ops.autotune(self.tune_descs, dev) or self.tune_descs, dev)
        self.graph: Optional[torch.cuda.CUDAGraph] = None

    def _build(self, net: "RefinementNet", P: int, H: int, W: int, with_posterior: bool, frames: int, packed: bool,
               A: "arena.Arena"):
        dev, lib = net.device, _lib.load()
        G = frames
        PF = P if packed else P    # boxes per frame (packed: the whole slot range may belong to one frame)
        P = P if packed else G * PF    # batch of the network body
        self.layout: List[tuple] = []  # packed: (frame, first slot, boxes) of the current call
        PK, DW = net.packed, net.packed_dw
        steps: List = []
        self.flops: Dict[str, float] = {}
        self.descs: List = []
        self.dw_bytes: Dict[str, float] = {}

        def alloc(n, h, w, c) -> NHWC:
            ps = (c + 3) // 4 * 4
            return NHWC(A.alloc(n, h, w, ps, "f32", pooled=ps == c), c=c)

        def release(v: Optional[NHWC]):                         # the last launch that reads ``v`` has been appended
            if v is not None:
                A.release(v)

        # bf16x3 mode (round 4): the depthwise half of a separable conv stores its result in the resident split layout S8 (in place of
        # the floats: {hi8, lo8} per group of 8 channels) and the pointwise half runs on csrc/conv_bf16x3_s8.hip, which stages it by
        # LDS-DMA; ASPP's concat buffer is S8 as well (its only reader is concat_projection)
        split_pw = net.precision == "bf16x3" and os.environ.get("PREMVOS_BF16X3_SPLIT", "1") != "0"
        self.split_pw = split_pw
        S8 = net.packed_s8
        self.tune_descs: List = []     # the launches premvos_conv2d_f32 runs (ops.autotune configures these; S8 convs have one kernel)

        def alloc_s8(n, h, w, c) -> NHWC:                       # (S8 buffers live in an arena of their own)
            assert c % 8 == 0
            return NHWC(A.alloc(n, h, w, c, "s8"), c=c, layout="s8")

        def conv(x, name, out, out_s8=None, **kw):
            """``x`` in S8 -> the S8 kernel (fp32 ``out`` and / or S8 ``out_s8``); fp32 ``x`` -> premvos_conv2d_f32."""
            key = f"conv:{name}"
            if x.layout == "s8":
                pk = S8[name]
                d = ops.conv_s8_desc(x, pk, out, out_s8, **kw)
                d.tile_hint = ops.S8_HINT
                o = out if out is not None else out_s8
                steps.append((key, lambda d=d, x=x, pk=pk, o8=out_s8: ops.run_s8(d, x, pk, o8)))
            else:
                assert out_s8 is None
                pk = PK[name]
                d = ops.conv_desc(x, pk, out, **kw)
                self.tune_descs.append(d)
                o = out
                steps.append((key, lambda d=d: ops.run_desc(d)))
            self.descs.append(d)
            self.flops[key] = 2.0 * o.n * o.h * o.w * pk.kh * pk.kw * pk.cin * pk.cout

        def dwconv(x: NHWC, name: str, out: NHWC, stride=1, rate=1, pre_relu=False, act=ACT_NONE):
            k = DW[name]
            assert x.c == k.c and out.c == k.c and x.layout == "f32"
            flags = act | (_lib.ACT_SPLIT8_BF16 if out.layout == "s8" else 0)

            def f(x=x, out=out, k=k):
                _lib.check(lib.premvos_dwconv3x3_f32(x.ptr, x.ps, x.n, x.h, x.w, x.c, k.wgt.data_ptr(),
                                                     k.bias.data_ptr(), k.c_pad, out.ptr, out.ps, out.h, out.w, stride,
                                                     rate, rate, rate, int(pre_relu), flags, _lib.current_stream()),
                           "dwconv3x3")
            steps.append((f"dw:{name}", f))
            self.dw_bytes[f"dw:{name}"] = 4.0 * k.c * (x.n * x.h * x.w + out.n * out.h * out.w)

        def alloc_mid(n, h, w, c, name=None):                   # the tensor between the two halves of a separable conv: S8 when its
            return alloc_s8(n, h, w, c) if split_pw and (name is None or name in S8) else alloc(n, h, w, c)   # pointwise half runs on the S8 kernel

        S = INPUT_SIZE
        self.frames = A.side((G, H, W, 3), torch.uint8)
        NG = 1 if packed else G        # leading dimension of the per-frame result views ([1, slots, ...] when packed)
        self.boxes_g = A.side((NG, PF, 4), torch.float32)      # y0 x0 y1 x1
        self.count = A.side((G,), torch.int32)
        self.crops = A.side((NG, PF, 4), torch.int32)
        self.frame, self.boxes = self.frames[0], self.boxes_g[0]                      # single-frame views
        self.net_in = alloc(P, S, S, 4)

        def mk_input():
            if packed:
                for g, off, n in self.layout:
                    _lib.check(lib.premvos_refine_input_u8(self.frames[g].data_ptr(), H, W, self.boxes_g[0, off:].data_ptr(),
                                                           self.count[g:].data_ptr(), n, S, self.net_in.images(off, n).ptr,
                                                           self.crops[0, off:].data_ptr(), _lib.current_stream()), "refine_input")
                return
            for g in range(G):
                _lib.check(lib.premvos_refine_input_u8(self.frames[g].data_ptr(), H, W, self.boxes_g[g].data_ptr(),
                                                       self.count[g:].data_ptr(), PF, S,
                                                       self.net_in.images(g * PF, PF).ptr, self.crops[g].data_ptr(),
                                                       _lib.current_stream()), "refine_input")
        steps.append(("refine_input", mk_input))

        # stem: conv2d_same 3x3 s2 (pad 1 + VALID) and 3x3 s1  (core/xception.py:430-433)
        h1 = ops.out_size(S, 3, 2, 1, 1)
        c11 = alloc(P, h1, h1, 32)
        conv(self.net_in, "entry_flow/conv1_1", c11, stride=(2, 2), pad=(1, 1), act=ACT_RELU)
        x = alloc(P, h1, h1, 64)
        conv(c11, "entry_flow/conv1_2", x, pad=(1, 1), act=ACT_RELU)
        release(c11)

        skip_feat: Optional[NHWC] = None
        mods = module_plan(net.num_middle)
        x8: Optional[NHWC] = None          # S8 copy of x (bf16x3 mode) when the next consumer is a conv: a shortcut conv / ASPP's 1x1 branch
        for mi_, (prefix, cin, depths, skip, relu_in, stride, rate) in enumerate(mods):
            inp, cur = x, x
            act = ACT_RELU if relu_in else ACT_NONE
            sc = None
            if skip == "conv":
                ho = ops.out_size(inp.h, 1, stride, 0, 0)
                sc = alloc(P, ho, ho, depths[-1])
                on_s8 = split_pw and prefix + "/shortcut" in S8
                if on_s8 and x8 is None:               # (its producer ran on an fp32 kernel: one split pass)
                    x8 = alloc_s8(P, inp.h, inp.w, inp.c)
                    steps.append((f"split8:{prefix}", lambda i_=inp, o_=x8: ops.split8(i_, o_)))
                conv(x8 if on_s8 else inp, prefix + "/shortcut", sc, stride=(stride, stride))
            if x8 is not None:
                release(x8)
                x8 = None
            # does the NEXT consumer of this module's output multiply it (shortcut conv of the next module, aspp0 after the last)?
            want8 = split_pw and ((mods[mi_ + 1][3] == "conv" and mods[mi_ + 1][0] + "/shortcut" in S8) if mi_ + 1 < len(mods) else True)
            for i, d in enumerate(depths):
                s = stride if i == 2 else 1
                ho = ops.out_size(cur.h, 3, s, rate, rate, rate)
                t = alloc_mid(P, ho, ho, cur.c, f"{prefix}/separable_conv{i + 1}_pointwise")
                dwconv(cur, f"{prefix}/separable_conv{i + 1}_depthwise", t, stride=s, rate=rate, pre_relu=not relu_in,
                       act=act)
                o = alloc(P, ho, ho, d)
                res = None
                if i == 2 and skip == "conv":
                    res = sc
                elif i == 2 and skip == "sum":
                    res = inp
                if i == 2 and want8 and t.layout == "s8":
                    x8 = alloc_s8(P, ho, ho, d)
                conv(t, f"{prefix}/separable_conv{i + 1}_pointwise", o, out_s8=x8 if i == 2 else None, act=act, res=res)
                release(t)
                if cur is not inp:
                    if f"{prefix}/separable_conv{i}" == DECODER_SKIP:
                        skip_feat = cur          # decoder end point: keep it alive
                    else:
                        release(cur)
                cur = o
            if sc is not None:
                release(sc)
            release(inp)
            x = cur
        feat = x
        self.xception_out = feat
        fh = feat.h

        # ASPP (model.py:383-433): [image pooling | 1x1 | 3 atrous separable] -> concat 1280 -> 1x1 256
        cat = alloc_mid(P, fh, fh, 1280)
        cat8 = cat.layout == "s8"
        gp = alloc(P, 1, 1, 2048)
        steps.append(("gap", lambda i=feat, o=gp: _lib.check(lib.premvos_global_avgpool_f32(
            i.ptr, i.ps, i.n, i.h * i.w, i.c, o.ptr, o.ps, _lib.current_stream()), "gap")))
        ip = alloc(P, 1, 1, 256)
        conv(gp, "image_pooling", ip, act=ACT_RELU)
        release(gp)
        bc = alloc(P, fh, fh, 256) if cat8 else cat.slice(0, 256)
        steps.append(("broadcast", lambda i=ip, o=bc: _lib.check(lib.premvos_broadcast_pixel_f32(
            i.ptr, i.ps, i.n, 256, o.ptr, o.ps, o.h, o.w, _lib.current_stream()), "broadcast")))
        release(ip)
        if cat8:       # the S8 concat buffer: the broadcast and the (fp32-input) 1x1 branch go through an fp32 block and are split once
            steps.append(("split8:image_pooling", lambda i=bc, o=cat.slice(0, 256): ops.split8(i, o)))
            release(bc)
            conv(x8, "aspp0", None, cat.slice(256, 256), act=ACT_RELU)        # (x8: the S8 copy exit_flow's last conv wrote)
            release(x8)
        else:
            conv(feat, "aspp0", cat.slice(256, 256), act=ACT_RELU)
        for i, r in enumerate(ATROUS_RATES, 1):
            t = alloc_mid(P, fh, fh, 2048)
            dwconv(feat, f"aspp{i}_depthwise", t, rate=r, act=ACT_RELU)
            if cat8:
                conv(t, f"aspp{i}_pointwise", None, cat.slice(256 * (i + 1), 256), act=ACT_RELU)
            else:
                conv(t, f"aspp{i}_pointwise", cat.slice(256 * (i + 1), 256), act=ACT_RELU)
            release(t)
        aspp = alloc(P, fh, fh, 256)
        conv(cat, "concat_projection", aspp, act=ACT_RELU)
        release(cat)
        self.aspp_out = aspp           # (named outputs -- xception_out, aspp_out, decoder_out, logits -- stay to the end of the list)

        # decoder (model.py:503-598): [aspp up-sampled (align_corners) | 1x1(skip) 48] -> 2 separable convs -> logits
        dh = int((float(S) - 1.0) * 0.25 + 1.0)                                   # scale_dimension
        assert skip_feat is not None and skip_feat.h == dh
        dcat = alloc(P, dh, dh, 304)
        steps.append(("resize_aspp", lambda i=aspp, o=dcat.slice(0, 256): _lib.check(lib.premvos_resize_bilinear_f32(
            i.ptr, i.ps, i.n, i.h, i.w, 256, o.ptr, o.ps, o.h, o.w, 1, _lib.current_stream()), "resize")))
        conv(skip_feat, "decoder/feature_projection0", dcat.slice(256, 48), act=ACT_RELU)
        release(skip_feat)
        d = dcat
        for j in (0, 1):
            t = alloc_mid(P, dh, dh, d.c, f"decoder/decoder_conv{j}_pointwise")
            dwconv(d, f"decoder/decoder_conv{j}_depthwise", t, act=ACT_RELU)
            release(d)
            o = alloc(P, dh, dh, 256)
            conv(t, f"decoder/decoder_conv{j}_pointwise", o, act=ACT_RELU)
            release(t)
            d = o
        self.decoder_out = d
        self.logits = alloc(P, dh, dh, 2)
        conv(d, "logits/features", self.logits)

        # SegmentationSoftmax eval branch + conf_score
        self.mask_g = A.side((NG, PF, H, W), torch.uint8)
        self.posterior_g = A.side((NG, PF, H, W), torch.float32) if with_posterior else None
        self.conf_g = A.side((NG, PF), torch.float32)
        self.mask, self.conf = self.mask_g[0], self.conf_g[0]                         # single-frame views
        self.posterior = self.posterior_g[0] if with_posterior else None
        wsb = int(lib.premvos_refine_output_workspace_bytes(PF, S, H, W))
        self.ws = A.side((wsb + 15) // 16 * 4, torch.float32)

        def out_layer(lg=self.logits):
            if packed:
                for g, off, n in self.layout:
                    lgf = lg.images(off, n)
                    _lib.check(lib.premvos_refine_output_f32(
                        lgf.ptr, lgf.ps, lgf.h, lgf.w, self.crops[0, off:].data_ptr(), self.count[g:].data_ptr(), n, S, H, W,
                        self.mask_g[0, off:].data_ptr(), self.posterior_g[0, off:].data_ptr() if with_posterior else None,
                        self.conf_g[0, off:].data_ptr(), self.ws.data_ptr(), _lib.current_stream()), "refine_output")
                return
            for g in range(G):          # stream-ordered, so the frames can share the scratch buffer
                lgf = lg.images(g * PF, PF)
                _lib.check(lib.premvos_refine_output_f32(
                    lgf.ptr, lgf.ps, lgf.h, lgf.w, self.crops[g].data_ptr(), self.count[g:].data_ptr(), PF, S, H, W,
                    self.mask_g[g].data_ptr(), self.posterior_g[g].data_ptr() if with_posterior else None,
                    self.conf_g[g].data_ptr(), self.ws.data_ptr(), _lib.current_stream()), "refine_output")
        steps.append(("refine_output", out_layer))
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


class RefinementNet:
    """``weights``: slim variable scopes (backbone keys without the 'xception_65/' prefix) ->
    '<scope>/weights' OIHW, '<scope>/depthwise_weights' [C,1,3,3], '<scope>/BatchNorm' = dict(gamma,beta,mean,var),
    'logits/features/biases'."""

    def __init__(self, weights: Dict[str, object], num_middle: int = 16, device=None, use_graph: bool = True,
                 precision: Optional[str] = None):
        _lib.require_gpu()
        self.precision = prec = precision or ops.default_precision()
        self.device, self.use_graph, self.num_middle = _lib.resolve_device(device), use_graph, num_middle
        device = self.device
        self.packed: Dict[str, ops.PackedConv] = {}
        self.packed_s8: Dict[str, ops.PackedConvS8] = {}     # bf16x3 mode: the convs whose input is resident in the S8 layout
        self.packed_dw: Dict[str, PackedDW] = {}
        self._plans: Dict[tuple, _Plan] = {}
        self._lane_bytes: Dict[int, dict] = {}       # lane -> {arena kind: the activation bytes its plans share}
        self._plans_lock = __import__("threading").Lock()
        self.max_plans = int(os.environ.get("PREMVOS_REFINE_MAX_PLANS", "12"))
        head = ("image_pooling", "aspp", "concat_projection", "decoder/")
        for k, v in weights.items():
            scope = k.rsplit("/", 1)[0]
            eps = EPS_HEAD if scope.startswith(head) else EPS_BACKBONE
            if k.endswith("/depthwise_weights"):
                self.packed_dw[scope] = PackedDW(v, weights[scope + "/BatchNorm"], eps, device)
            elif k.endswith("/weights"):
                if scope + "/BatchNorm" in weights:
                    scale, bias = _fold(weights[scope + "/BatchNorm"], eps)
                    # bf16x3 mode: the 1x1 convs with K >= PREMVOS_S8_MIN_CIN (256) input channels run on the S8 kernel; the short-K
                    # layers of the entry flow are HBM-bound (the fp32 streaming kernel moves them at 4.7 TB/s: 0.98 ms for the 64 -> 128
                    # layer at 193 x 193 against 2.4 ms on a 256-row MFMA tile with two K stages) and, like the dense 3x3 stem,
                    # stay on the fp32 kernels the shipped table tunes
                    on_s8 = prec == "bf16x3" and os.environ.get("PREMVOS_BF16X3_SPLIT", "1") != "0"
                    s8 = on_s8 and v.shape[1] >= int(os.environ.get("PREMVOS_S8_MIN_CIN", "256")) and v.shape[0] % 8 == 0 and \
                        (scope.endswith("_pointwise") or scope.endswith("/shortcut") or scope in ("concat_projection", "aspp0"))
                    if s8:
                        self.packed_s8[scope] = ops.pack_conv_s8(v, bias, device, scale=scale)
                        continue
                    self.packed[scope] = ops.pack_conv(v, bias, device, scale=scale, precision="fp32" if on_s8 else prec)
                else:
                    on_s8 = prec == "bf16x3" and os.environ.get("PREMVOS_BF16X3_SPLIT", "1") != "0"
                    self.packed[scope] = ops.pack_conv(v, weights.get(scope + "/biases"), device, precision="fp32" if on_s8 else prec)

    def plan(self, P: int, H: int, W: int, with_posterior: bool = False, lane: int = 0, frames: int = 1,
             packed: bool = False) -> _Plan:
        """``lane`` selects an independent workspace (same weights) so several calls can be in flight; ``frames`` > 1
        builds a plan that refines that many frames (P boxes each) as one batch; ``packed``: P slots shared by <= frames frames."""
        key = (P, H, W, with_posterior, lane, frames, packed)
        with self._plans_lock:                    # LRU: a hit moves the plan to the young end (dicts keep insertion order)
            p = self._plans.pop(key, None)
            if p is not None:
                self._plans[key] = p
                return p
        with ops.BUILD_LOCK:                      # lanes of the file drivers run on several threads; building + tuning +
            with self._plans_lock:                # graph capture of a plan is done by one of them at a time
                p = self._plans.get(key)
            if p is None:
                p = _Plan(self, P, H, W, with_posterior, frames, packed, lane)
                if self.use_graph and not packed:
                    torch.cuda.synchronize()      # capture must not race kernels of another lane's stream
                    p.capture()
                with self._plans_lock:
                    self._plans[key] = p
                    # every plan holds a full Xception activation set: evict the least recently used one OF THIS LANE (another lane's
                    # plan may be executing on its own thread right now) -- the bound is PER LANE (max_plans each): two lanes over varying proposal counts must not evict, rebuild and
                    # re-tune each other's activation sets
                    while True:
                        mine = [k for k in self._plans if k[4] == lane and k != key]
                        if len(mine) < self.max_plans:
                            break
                        self._plans.pop(mine[0])
        return p

    def refine(self, frame_rgb: torch.Tensor, boxes_y0x0y1x1: torch.Tensor, max_boxes: Optional[int] = None,
               with_posterior: bool = False, lane: int = 0) -> _Plan:
        """frame uint8 [H,W,3] RGB, boxes float [n,4] (y0,x0,y1,x1).  Results stay in the returned plan:
        ``mask`` uint8 [P,H,W], ``conf`` [P], ``posterior`` (optional), valid for the first n entries."""
        n = boxes_y0x0y1x1.shape[0]
        P = max_boxes or max(n, 1)
        assert n <= P
        H, W, _ = frame_rgb.shape
        p = self.plan(P, H, W, with_posterior, lane)
        p.frame.copy_(frame_rgb)
        p.boxes.zero_()
        if n:
            p.boxes[:n].copy_(boxes_y0x0y1x1)
        p.count.fill_(n)
        if p.graph is not None:
            p.graph.replay()
        else:
            p.run()
        return p

    def refine_packed(self, frames_rgb: torch.Tensor, boxes_per_frame: List[torch.Tensor], slots: int, max_frames: int,
                      lane: int = 0) -> _Plan:
        """Several frames at once with the boxes PACKED: frames uint8 [g,H,W,3] (g <= max_frames), ``boxes_per_frame[i]`` float
        [n_i,4] (host or device), sum(n_i) <= ``slots``.  Results in the plan: ``mask_g[0]`` [slots,H,W], ``conf_g[0]`` [slots];
        the boxes of frame i occupy the slots [sum(n_<i), sum(n_<=i)).  Same numbers as ``refine`` on each frame (every crop is
        an independent batch element); the launches are eager (slot offsets are launch arguments)."""
        g, H, W, _ = frames_rgb.shape
        counts = [int(b.shape[0]) for b in boxes_per_frame]
        total = sum(counts)
        assert g == len(counts) <= max_frames and total <= slots
        p = self.plan(slots, H, W, False, lane, frames=max_frames, packed=True)
        p.frames[:g].copy_(frames_rgb)
        p.boxes_g[0].zero_()
        if total:
            p.boxes_g[0, :total].copy_(torch.cat([torch.as_tensor(b, dtype=torch.float32).reshape(-1, 4) for b in boxes_per_frame]))
        cnt = torch.zeros((max_frames,), dtype=torch.int32)
        cnt[:g] = torch.tensor(counts, dtype=torch.int32)
        p.count.copy_(cnt)
        offs = [sum(counts[:i]) for i in range(g)]
        p.layout = [(i, offs[i], counts[i]) for i in range(g) if counts[i] > 0]
        p.run()                             # (unused slots run on whatever finite crops an earlier call left there; never read)
        return p

    def refine_group(self, frames_rgb: torch.Tensor, boxes_y0x0y1x1: torch.Tensor, counts: Optional[torch.Tensor] = None,
                     with_posterior: bool = False, lane: int = 0) -> _Plan:
        """Several frames at once: frames uint8 [G,H,W,3], boxes float [G,P,4], counts int32 [G] (default: all P valid).
        Results in the plan: ``mask_g`` [G,P,H,W], ``conf_g`` [G,P], ``posterior_g``.  Same numbers as G ``refine``
        calls (each crop is an independent batch element)."""
        G, H, W, _ = frames_rgb.shape
        assert boxes_y0x0y1x1.shape[0] == G and boxes_y0x0y1x1.shape[2] == 4
        P = boxes_y0x0y1x1.shape[1]
        p = self.plan(P, H, W, with_posterior, lane, frames=G)
        p.frames.copy_(frames_rgb)
        p.boxes_g.copy_(boxes_y0x0y1x1)
        if counts is None:
            p.count.fill_(P)
        else:
            p.count.copy_(counts)
        if p.graph is not None:
            p.graph.replay()
        else:
            p.run()
        return p
