"""PWC-DC-Net forward on MI355X: the host-side mirror of the reference module interface.

Reference interface kept (optical_flow_net-PWC-Net/models/PWCNet.py):
    net = pwc_dc_net(path)            # :496-505  (torch pickle, optional 'state_dict' wrapper)
    net = net.cuda(); net.eval()
    flow2 = net(x)                    # x [B,6,H,W] fp32 (BGR/255, two frames) -> [B,2,H/4,W/4]  :179-272

Design (not a translation): activations live in NHWC; both frames of a pair go through the shared
weight pyramid as one batch; every DenseNet estimator level owns ONE pre-allocated concat buffer and
each conv writes its output into the channel window the reference's ``torch.cat((conv(x), x), 1)``
would have produced (:201-205) -- no concat copies; the cost-volume kernel writes LeakyReLU(corr) and
the c1 features straight into that buffer; the transposed convs are 3x3 convs with a pixel-shuffle
epilogue writing into the next level's buffer; the residual ``flow2 += dc_conv7(...)`` is a conv
epilogue.  The whole forward is a fixed list of kernel launches built once per input shape
(``_Plan``) so it can be replayed from a captured HIP graph.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from .. import _lib, ops
from ..ops import ACT_LEAKY, ACT_NONE, NHWC

MD = 4
ND = (2 * MD + 1) ** 2
GROWTH = (128, 128, 96, 64, 32)
GROW_SUM = sum(GROWTH)                                  # 448
FEAT = {1: 16, 2: 32, 3: 64, 4: 96, 5: 128, 6: 196}
PYR = {1: ("conv1a", "conv1aa", "conv1b"), 2: ("conv2a", "conv2aa", "conv2b"),
       3: ("conv3a", "conv3aa", "conv3b"), 4: ("conv4a", "conv4aa", "conv4b"),
       5: ("conv5a", "conv5aa", "conv5b"), 6: ("conv6aa", "conv6a", "conv6b")}
FLOW_SCALE = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}
CONTEXT = (("dc_conv1", 128, 1), ("dc_conv2", 128, 2), ("dc_conv3", 128, 4), ("dc_conv4", 96, 8),
           ("dc_conv5", 64, 16), ("dc_conv6", 32, 1))


def _od(level: int) -> int:
    return ND if level == 6 else ND + FEAT[level] + 4


FLOW_WINO4_MIN_C = int(os.environ.get("PREMVOS_FLOW_WINO4_MIN_C", "64"))


class _Plan:
    """Workspace + launch list for one (B, H, W)."""

    def __init__(self, net: "PWCDCNet", b: int, h: int, w: int):
        dev = net.device
        self.b, self.h, self.w = b, h, w
        self.x_in = torch.empty((b, 6, h, w), dtype=torch.float32, device=dev)
        self.flow_out = torch.empty((b, 2, h // 4, w // 4), dtype=torch.float32, device=dev)
        steps: List = []
        P = net.packed
        keep: List[NHWC] = []          # descriptors hold raw pointers: every buffer must outlive the plan

        def alloc(*a):
            v = NHWC.alloc(*a)
            keep.append(v)
            return v

        self.flops: Dict[str, float] = {}   # algorithmic FLOPs (2*MAC, true cin/cout) per conv launch
        self.descs: List = []
        dense: Dict[int, List] = {}         # estimator level -> [(step index, desc, first channel of the window in the level's buffer)]

        def conv(x, name, out, **kw):
            pk = P[name]
            d = ops.conv_desc(x, pk, out, **kw)
            self.descs.append(d)
            steps.append(("conv:" + name, lambda d=d: ops.run_desc(d)))
            if pk.cout_ps:   # transposed conv k4 s2: every input pixel feeds 16 taps
                self.flops["conv:" + name] = 2.0 * x.n * x.h * x.w * 16 * pk.cin * pk.cout_ps
            else:
                self.flops["conv:" + name] = 2.0 * out.n * out.h * out.w * pk.kh * pk.kw * pk.cin * pk.cout

        # frames -> NHWC [2B,H,W,4]: images [0,B) = frame 1, [B,2B) = frame 2
        img = alloc(2 * b, h, w, 3, dev)
        for i in range(b):
            for f in range(2):
                src = self.x_in[i, 3 * f:3 * f + 3].unsqueeze(0)
                dst = img.images(f * b + i, 1)
                steps.append(("nchw_to_nhwc", lambda s=src, d=dst: ops.nchw_to_nhwc(s, d)))
        self.n_pre = len(steps)
        self.img = img

        # siamese pyramid (PWCNet.py:183-194), 2B images per launch
        feats: Dict[int, NHWC] = {}
        cur, ch, cw = img, h, w
        for lvl in range(1, 7):
            ch, cw = ch // 2, cw // 2
            a, aa, bb = PYR[lvl]
            t1 = alloc(2 * b, ch, cw, FEAT[lvl], dev)
            t2 = alloc(2 * b, ch, cw, FEAT[lvl], dev)
            t3 = alloc(2 * b, ch, cw, FEAT[lvl], dev)
            conv(cur, a, t1, stride=(2, 2), pad=(1, 1), act=ACT_LEAKY)
            conv(t1, aa, t2, pad=(1, 1), act=ACT_LEAKY)
            conv(t2, bb, t3, pad=(1, 1), act=ACT_LEAKY)
            feats[lvl] = cur = t3

        # coarse-to-fine estimators (PWCNet.py:197-264)
        self.level_flow: Dict[int, NHWC] = {}
        prev_x: Optional[NHWC] = None
        xbufs: Dict[int, NHWC] = {}
        for lvl in (6, 5, 4, 3, 2):
            lh, lw, fc = h >> lvl, w >> lvl, FEAT[lvl]
            od = _od(lvl)
            xbufs[lvl] = alloc(b, lh, lw, GROW_SUM + od, dev)
        for lvl in (6, 5, 4, 3, 2):
            lh, lw, fc = h >> lvl, w >> lvl, FEAT[lvl]
            od = _od(lvl)
            X = xbufs[lvl]
            c1 = feats[lvl].images(0, b)
            c2 = feats[lvl].images(b, b)
            copy_f1 = lvl != 6
            dst = X.slice(GROW_SUM, ND + (fc if copy_f1 else 0))
            if lvl == 6:
                f2 = c2
            else:
                # warp (PWCNet.py:207,220,233,246) then cost volume: the fused premvos_warp_corr_fwd_f32 is bit-identical but
                # measured slower (the 8x32 tile's halo repeats the bilinear gathers 2.5x), so the warped map is materialised
                up_flow = X.slice(GROW_SUM + ND + fc, 2)       # written by the level above
                wbuf = alloc(b, lh, lw, fc, dev)
                steps.append((f"warp{lvl}", lambda x=c2, f=up_flow, s=FLOW_SCALE[lvl], o=wbuf: ops.warp(x, f, s, o)))
                f2 = wbuf
            steps.append((f"corr{lvl}", lambda a=c1, bq=f2, o=dst, cp=copy_f1: ops.corr(a, bq, o, MD, 0.1, cp)))
            off = GROW_SUM
            for i, g in enumerate(GROWTH):
                xin = X.slice(off, GROW_SUM + od - off)
                conv(xin, f"conv{lvl}_{i}", X.slice(off - g, g), pad=(1, 1), act=ACT_LEAKY)
                dense.setdefault(lvl, []).append((len(steps) - 1, self.descs[-1], off))
                off -= g
            flow = alloc(b, lh, lw, 2, dev)               # ps = 4 (pad lanes stay zero)
            self.level_flow[lvl] = flow
            conv(X, f"predict_flow{lvl}", flow, pad=(1, 1))
            if lvl != 2:
                nfc = FEAT[lvl - 1]
                nx = xbufs[lvl - 1]
                conv(flow, f"deconv{lvl}", nx.slice(GROW_SUM + ND + nfc, 2), pad=(1, 1))
                conv(X, f"upfeat{lvl}", nx.slice(GROW_SUM + ND + nfc + 2, 2), pad=(1, 1))
            prev_x = X

        # context network (PWCNet.py:266-267) + residual
        y = prev_x
        lh, lw = h >> 2, w >> 2
        for name, co, dil in CONTEXT:
            o = alloc(b, lh, lw, co, dev)
            conv(y, name, o, pad=(dil, dil), dilation=(dil, dil), act=ACT_LEAKY)
            y = o
        flow2 = alloc(b, lh, lw, 2, dev)
        conv(y, "dc_conv7", flow2, pad=(1, 1), res=self.level_flow[2])
        self.flow2_nhwc = flow2
        self.n_core_end = len(steps)
        steps.append(("nhwc_to_nchw", lambda s=flow2, d=self.flow_out: ops.nhwc_to_nchw(s, d)))
        self.steps = steps
        self.buffers = keep
        self.ws = ops.assign_workspace(ops.autotune(self.descs, dev) or self.descs, dev)      # split-K scratch shared by the whole launch list
        # Kept Winograd slabs (round 4): the F(4x4) layers of an estimator level transform only the channels the previous layer
        # added to the level's concat buffer (PWCNet.py:201-205 prepends them); one slab, reused level after level.  Same bits.
        self.vslab, self.vslab_layers = None, {}
        if ops.vslab_enabled():
            plans = {}
            for lvl, lst in dense.items():
                hints = [d.tile_hint == 4 for _, d, _ in lst]
                if True not in hints:
                    continue
                first = hints.index(True)
                stop = hints.index(False, first) if False in hints[first:] else len(lst)         # the first consecutive run of F(4x4) layers
                if stop - first >= 2:                                                            # (a single layer has nothing to share)
                    pl = ops.wino4_slab_plan([(d, off) for _, d, off in lst[first:stop]])
                    if pl is not None:
                        plans[lvl] = (lst[first:stop], pl)
            if plans:
                self.vslab = torch.empty(max(pl[0] for _, pl in plans.values()), dtype=torch.float32, device=dev)
                for lvl, (run, (_, plan)) in plans.items():
                    for (idx, _, _), (d, pitch, c0, t_cn) in zip(run, plan):
                        steps[idx] = (steps[idx][0], lambda d=d, p=pitch, c=c0, t=t_cn: ops.run_wino4_slab(d, self.vslab, p, c, t))
                        self.vslab_layers[steps[idx][0]] = (c0, t_cn)
        self.feats, self.xbufs = feats, xbufs
        self.graph: Optional[torch.cuda.CUDAGraph] = None

    @property
    def core_steps(self):
        """The network proper: NHWC frames in ``self.img`` -> ``self.flow2_nhwc``."""
        return self.steps[self.n_pre:self.n_core_end]

    def run(self, steps=None):
        for _, fn in (self.steps if steps is None else steps):
            fn()

    def capture(self, steps=None) -> "torch.cuda.CUDAGraph":
        """Record a launch list into a HIP graph (launch-bound coarse levels replay as one submit)."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.run(steps)     # warm-up outside capture
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):      # other host threads (IO lanes) keep using the GPU
            self.run(steps)
        if steps is None:
            self.graph = g
        return g


class PWCDCNet:
    """Drop-in for ``models.PWCNet.PWCDCNet`` (inference).  ``__call__`` == ``forward`` in eval mode."""

    def __init__(self, md: int = 4, device=None, use_graph: bool = True, precision: Optional[str] = None):
        assert md == MD, "PWC-Net is instantiated with md=4 (PWCNet.py:43)"
        self.precision = precision or ops.default_precision()
        self.device = _lib.resolve_device(device)
        self.use_graph = use_graph
        self.training = False
        self.packed: Dict[str, ops.PackedConv] = {}
        self._plans: Dict[tuple, _Plan] = {}

    # -- nn.Module-compatible surface the reference driver touches (script_pwc_multi.py:88-90) --
    def cuda(self):
        return self

    def eval(self):
        self.training = False
        return self

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        _lib.require_gpu()
        packed = {}
        names = {k.rsplit(".", 1)[0] for k in sd}
        for full in sorted(names):
            w, bia = sd[full + ".weight"], sd.get(full + ".bias")
            key = full[:-2] if full.endswith(".0") else full
            if key.startswith(("deconv", "upfeat")):
                packed[key] = ops.pack_deconv4x4s2(w, bia, self.device, self.precision)
            else:
                # F(4x4,3x3) from 64 input channels here (the other nets: 128): conv2_0 757 -> 600 us, dc_conv6 407 -> 335, conv3aa / conv3b
                # 167 -> 139 (tools/dev/cands_3x3.py); its ~1e-5 rounding is two orders below the flow's 1e-3 px bar
                # (the stride-2 pyramid layers conv*a cannot run it: they keep the default, and with it their table signatures)
                strided = key in {PYR[lvl][0] for lvl in PYR}
                packed[key] = ops.pack_conv(w, bia, self.device, precision=self.precision, wino4_min_c=None if strided else FLOW_WINO4_MIN_C)
        if strict:
            need = set()
            for lv in PYR.values():
                need.update(lv)
            for lvl in (6, 5, 4, 3, 2):
                need.update(f"conv{lvl}_{i}" for i in range(5))
                need.add(f"predict_flow{lvl}")
                if lvl != 2:
                    need.update((f"deconv{lvl}", f"upfeat{lvl}"))
            need.update(n for n, _, _ in CONTEXT)
            need.add("dc_conv7")
            missing = need - set(packed)
            if missing:
                raise KeyError(f"missing keys in state_dict: {sorted(missing)}")
        self.packed = packed
        self._plans.clear()
        return self

    def plan(self, b: int, h: int, w: int) -> _Plan:
        key = (b, h, w)
        if key not in self._plans:
            if h % 64 or w % 64:
                raise ValueError("PWC-Net input must be a multiple of 64 (script_pwc_multi.py:38-45)")
            p = _Plan(self, b, h, w)
            if self.use_graph:
                p.capture()
            self._plans[key] = p
        return self._plans[key]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _lib.require_gpu()
        assert x.dim() == 4 and x.shape[1] == 6 and x.dtype == torch.float32 and x.is_cuda
        b, _, h, w = x.shape
        p = self.plan(b, h, w)
        p.x_in.copy_(x)
        if p.graph is not None:
            p.graph.replay()
        else:
            p.run()
        return p.flow_out.clone()

    __call__ = forward


def pwc_dc_net(path: Optional[str] = None, device=None, use_graph: bool = True,
               precision: Optional[str] = None) -> PWCDCNet:
    """models/PWCNet.py:496-505."""
    model = PWCDCNet(device=device, use_graph=use_graph, precision=precision)
    if path is not None:
        data = torch.load(path, map_location="cpu")
        model.load_state_dict(data["state_dict"] if "state_dict" in data else data)
    return model
