"""ReID embedding net on the HIP path: a wide pre-activation ResNet on 128x128 box crops -> 128-d
embedding (code/ReID_net/configs/run:37-70; network/NetworkLayers.py:101-210,231-250; NetworkOutputLayers.py:253-272).

Mapping onto the kernels of this library:
  * every conv / FC is ``premvos_conv2d_f32`` (TF 'SAME' = asymmetric pad, extra pixel after);
  * ResidualUnit2 = BN0+ReLU on the unit input (``premvos_scale_shift_relu_f32``: the raw input is still needed by the
    identity shortcut) -> [1x1 shortcut conv on the ACTIVATED tensor when shape changes] -> conv W1 with the NEXT
    BatchNorm and ReLU folded into its epilogue -> ... -> last conv with the shortcut added in the epilogue;
  * conv1: BN+ReLU pass -> 3x3 conv -> SAME 3x3/3 max-pool; fc1/fc2/outputTriplet: BatchNorm on their input folded into
    the matrix (it is linear there), ReLU in the epilogue of fc1/fc2, the output layer is linear.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import _lib, ops
from ..ops import ACT_NONE, ACT_RELU, NHWC

BN_EPS = 1e-5                  # NetworkLayers.py:13
INPUT_SIZE = 128               # configs/run:27
CONTEXT = 1.2                  # configs/run:33-34
# configs/run:40-63: (name, features per conv, filter sizes, strides)
UNITS = ([("res0", (128, 128), (3, 3), (2, 1)), ("res1", (128, 128), (3, 3), (1, 1)), ("res2", (128, 128), (3, 3), (1, 1)),
          ("res3", (256, 256), (3, 3), (2, 1)), ("res4", (256, 256), (3, 3), (1, 1)), ("res5", (256, 256), (3, 3), (1, 1)),
          ("res6", (512, 512), (3, 3), (2, 1))]
         + [(f"res{i}", (512, 512), (3, 3), (1, 1)) for i in range(7, 12)]
         + [("res12", (512, 1024), (3, 3), (1, 2)), ("res13", (512, 1024), (3, 3), (1, 1)), ("res14", (512, 1024), (3, 3), (1, 1)),
            ("res15", (512, 1024, 2048), (1, 3, 1), (1, 2, 1)), ("res16", (1024, 2048, 4096), (1, 3, 1), (1, 1, 1))])


def _fold(bn: Dict[str, torch.Tensor]):
    scale = bn["gamma"].double() / torch.sqrt(bn["var"].double() + BN_EPS)
    shift = bn["beta"].double() - bn["mean"].double() * scale
    return scale.float(), shift.float()


def _same(size: int, k: int, s: int):
    out = -(-size // s)
    tot = max((out - 1) * s + k - size, 0)
    return out, tot // 2


def context_boxes(boxes_xywh, height: int, width: int, feed: bool = True) -> np.ndarray:
    """Context region x1.2, tf.round, clip -- in float32 like the TF graph (DAVIS_Forward_Feed.py:36-60 with an excess of
    at least one pixel when ``feed``; Similarity.py:267-287 otherwise) -> int32 [n,4] (x, y, w, h)."""
    b = np.asarray(boxes_xywh, np.float32).reshape(-1, 4).copy()
    xs, ys, ws, hs = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    f = np.float32(CONTEXT - 1.0)
    xs = xs - np.float32(0.5) * ws * f
    ys = ys - np.float32(0.5) * hs * f
    ws = ws * np.float32(CONTEXT)
    hs = hs * np.float32(CONTEXT)
    xs, ys, ws, hs = (np.rint(v).astype(np.int32) for v in (xs, ys, ws, hs))
    xs, ys = np.maximum(xs, 0), np.maximum(ys, 0)
    lo = 1 if feed else 0
    ws = ws - np.maximum(xs + ws - width, lo)
    hs = hs - np.maximum(ys + hs - height, lo)
    return np.ascontiguousarray(np.stack([xs, ys, ws, hs], 1).astype(np.int32))


class _Plan:
    def __init__(self, net: "ReIDNet", n: int, H: int, W: int, feed: bool):
        dev, lib = net.device, _lib.load()
        self.n, self.H, self.W = n, H, W
        steps: List = []
        keep: List = []
        self.flops: Dict[str, float] = {}
        self.descs: List = []
        PK = net.packed

        def alloc(nn, h, w, c):
            v = NHWC.alloc(nn, h, w, c, dev)
            keep.append(v)
            return v

        def conv(x, name, out, **kw):
            pk = PK[name]
            d = ops.conv_desc(x, pk, out, **kw)
            self.descs.append(d)
            key = f"conv:{name}"
            steps.append((key, lambda d=d: ops.run_desc(d)))
            self.flops[key] = 2.0 * out.n * out.h * out.w * pk.kh * pk.kw * pk.cin * pk.cout

        def conv_same(x, name, cout, stride=1, **kw):
            k = PK[name].kh
            ho, pt = _same(x.h, k, stride)
            wo, pl = _same(x.w, k, stride)
            out = alloc(x.n, ho, wo, cout)
            conv(x, name, out, stride=(stride, stride), pad=(pt, pl), **kw)
            return out

        def bn_relu(x, name):
            s, t = net.affine[name]
            out = alloc(x.n, x.h, x.w, x.c)
            steps.append((f"bn:{name}", lambda x=x, out=out, s=s, t=t: _lib.check(lib.premvos_scale_shift_relu_f32(
                x.ptr, x.ps, x.n * x.h * x.w, x.c, s.data_ptr(), t.data_ptr(), out.ptr, out.ps, 1, _lib.current_stream()),
                "scale_shift_relu")))
            return out

        S = INPUT_SIZE
        self.frame = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
        self.boxes = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        self.net_in = alloc(n, S, S, 3)                       # pixel stride 4, 4th channel 0
        steps.append(("reid_input", lambda: _lib.check(lib.premvos_reid_input_u8(
            self.frame.data_ptr(), H, W, self.boxes.data_ptr(), n, S, int(feed), self.net_in.ptr, _lib.current_stream()),
            "reid_input")))
        x = conv_same(self.net_in, "conv0/W", 64)
        self.unit_out: Dict[str, NHWC] = {}
        for name, feats, ks, st in net.units:
            a = bn_relu(x, f"{name}/bn0")
            sres = int(np.prod(st))
            res = conv_same(a, f"{name}/W0", feats[-1], sres) if f"{name}/W0" in PK else x
            cur = a
            for i in range(1, len(feats)):                   # BN(i+1) + ReLU folded into conv i
                cur = conv_same(cur, f"{name}/W{i}", feats[i - 1], st[i - 1], act=ACT_RELU)
            x = conv_same(cur, f"{name}/W{len(feats)}", feats[-1], st[-1], res=res)
            self.unit_out[name] = x
        a = bn_relu(x, "conv1/bn")
        y = conv_same(a, "conv1/W", PK["conv1/W"].cout)
        ho, pt = _same(y.h, 3, 3)
        wo, pl = _same(y.w, 3, 3)
        pooled = alloc(n, ho, wo, y.c)
        steps.append(("maxpool", lambda i=y, o=pooled: _lib.check(lib.premvos_maxpool_f32(
            i.ptr, i.ps, i.n, i.h, i.w, i.c, o.ptr, o.ps, o.h, o.w, 3, 3, pt, pl, float("-inf"), _lib.current_stream()),
            "maxpool")))
        self.pooled = pooled
        assert pooled.ps == pooled.c, "NHWC flatten needs an unpadded pixel stride"
        flat = NHWC(pooled.buf.view(n, 1, 1, ho * wo * pooled.c), c=ho * wo * pooled.c)
        h1 = alloc(n, 1, 1, PK["fc1/W"].cout)
        conv(flat, "fc1/W", h1, act=ACT_RELU)
        h2 = alloc(n, 1, 1, PK["fc2/W"].cout)
        conv(h1, "fc2/W", h2, act=ACT_RELU)
        self.emb = alloc(n, 1, 1, PK["outputTriplet/W"].cout)
        conv(h2, "outputTriplet/W", self.emb)
        self.steps, self.buffers = steps, keep
        self.ws = ops.assign_workspace(ops.autotune(self.descs, dev) or self.descs, dev)
        self.graph: Optional[torch.cuda.CUDAGraph] = None

    def run(self):
        for _, fn in self.steps:
            fn()

    def capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.run()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):      # other host threads (IO lanes) keep using the GPU
            self.run()
        self.graph = g
        return g

    @property
    def embeddings(self) -> torch.Tensor:
        e = self.emb
        return e.buf.view(e.n, -1)[:, :e.c]


class ReIDNet:
    """``weights``: TF variable names of the ReID graph -> '<layer>/W' conv OIHW | FC [out, in(NHWC-flattened)],
    '<layer>/b', '<layer>/bn*' = dict(gamma, beta, mean, var)  (mean_ema / var_ema in the checkpoint)."""

    def __init__(self, weights: Dict[str, object], device=None, use_graph: bool = True,
                 units: Sequence = UNITS, precision: Optional[str] = None):
        _lib.require_gpu()
        prec = precision or ops.default_precision()
        self.device, self.use_graph, self.units = _lib.resolve_device(device), use_graph, list(units)
        device = self.device
        self.packed: Dict[str, ops.PackedConv] = {}
        self.affine: Dict[str, tuple] = {}
        self._plans: Dict[tuple, _Plan] = {}
        W = weights
        self.packed["conv0/W"] = ops.pack_conv(W["conv0/W"], None, device, precision=prec)
        for name, feats, ks, st in self.units:
            s, t = _fold(W[f"{name}/bn0"])
            self.affine[f"{name}/bn0"] = (s.to(device), t.to(device))
            if f"{name}/W0" in W:
                self.packed[f"{name}/W0"] = ops.pack_conv(W[f"{name}/W0"], None, device, precision=prec)
            for i in range(1, len(feats) + 1):
                if i < len(feats):      # the BatchNorm that follows conv i (named bn{i+1}) goes into its epilogue
                    s, t = _fold(W[f"{name}/bn{i + 1}"])
                    self.packed[f"{name}/W{i}"] = ops.pack_conv(W[f"{name}/W{i}"], t, device, scale=s, precision=prec)
                else:
                    self.packed[f"{name}/W{i}"] = ops.pack_conv(W[f"{name}/W{i}"], None, device, precision=prec)
        s, t = _fold(W["conv1/bn"])
        self.affine["conv1/bn"] = (s.to(device), t.to(device))
        self.packed["conv1/W"] = ops.pack_conv(W["conv1/W"], None, device, precision=prec)
        for name in ("fc1", "fc2", "outputTriplet"):         # y = (x*s + t) @ W^T + b = x @ (W*s)^T + (W @ t + b)
            s, t = _fold(W[f"{name}/bn"])
            w = W[f"{name}/W"].float()
            self.packed[f"{name}/W"] = ops.pack_conv((w * s.view(1, -1)).view(w.shape[0], w.shape[1], 1, 1),
                                                     w.double().mv(t.double()).float() + W[f"{name}/b"].float(), device,
                                                     precision=prec)

    def plan(self, n: int, H: int, W: int, feed: bool = True) -> _Plan:
        key = (n, H, W, feed)
        if key not in self._plans:
            p = _Plan(self, n, H, W, feed)
            if self.use_graph:
                p.capture()
            self._plans[key] = p
        return self._plans[key]

    def embed(self, frame_rgb: torch.Tensor, boxes_xywh, max_boxes: Optional[int] = None, feed: bool = True) -> torch.Tensor:
        """frame uint8 [H,W,3] RGB (CUDA), boxes [n,4] xywh floats (host) -> embeddings float32 [n,128] (CUDA view, valid
        until the next call)."""
        H, W, _ = frame_rgb.shape
        cb = context_boxes(boxes_xywh, H, W, feed)
        n = len(cb)
        P = max_boxes or max(n, 1)
        assert 0 < n <= P
        p = self.plan(P, H, W, feed)
        p.frame.copy_(frame_rgb)
        full = np.zeros((P, 4), np.int32)
        full[:n] = cb
        p.boxes.copy_(torch.from_numpy(full))
        if p.graph is not None:
            p.graph.replay()
        else:
            p.run()
        return p.embeddings[:n]
