"""Eager stand-in for the slice of tf.contrib.slim (TF 1.8) that the reference's DeepLabv3+ / Xception-65 graph code calls
(refinement_net/network/deeplab/{model.py, core/xception.py, core/feature_extractor.py}), built on tools/tfshim.py's tensors.
Same contract as tfshim: the reference's composition (arg-scopes, paddings, strides -> atrous switch, scopes / variable names,
resize calls, concat order) runs as written; conv2d / separable_conv2d / batch_norm / resize_bilinear themselves are this
file's restatement of the published TF semantics (NHWC, TF 'SAME' = extra pad after, fused inference batch norm,
legacy / align_corners bilinear) and stay third-party.

Dev / test-generation tool only.
"""
from __future__ import annotations

import contextlib
import functools
import types
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

import tfshim
from tfshim import T, _np, get_variable, variable_scope

_STACK: List[Dict[str, dict]] = [{}]
LAYERS: List[dict] = []          # one record per conv / separable conv the graph code instantiated (scope, geometry, ...)


def _key(f):
    return getattr(f, "_slim_key", getattr(f, "__name__", str(f)))


@contextlib.contextmanager
def arg_scope(list_ops_or_scope, **kwargs):
    """slim.arg_scope([ops], **kw) adds defaults; slim.arg_scope(scope_dict) re-enters a captured scope."""
    if isinstance(list_ops_or_scope, dict):
        new = {k: dict(v) for k, v in list_ops_or_scope.items()}
    else:
        new = {k: dict(v) for k, v in _STACK[-1].items()}
        for op in list_ops_or_scope:
            new.setdefault(_key(op), {}).update(kwargs)
    _STACK.append(new)
    try:
        yield new
    finally:
        _STACK.pop()


def add_arg_scope(func):
    @functools.wraps(func)
    def wrapped(*args, **kwargs):
        merged = dict(_STACK[-1].get(_key(wrapped), {}))
        merged.update(kwargs)
        return func(*args, **merged)
    wrapped._slim_key = func.__module__ + "." + func.__name__
    return wrapped


def _same_pads(size, k_eff, s):
    out = -(-size // s)
    tot = max((out - 1) * s + k_eff - size, 0)
    return tot // 2, tot - tot // 2


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(_np(x), dtype=np.float32))


def _conv_nhwc(x, w_oihw, stride, rate, padding, groups=1):
    xt = _t(x).permute(0, 3, 1, 2)
    kh, kw = w_oihw.shape[2:]
    if padding.upper() == "SAME":
        pt, pb = _same_pads(xt.shape[2], (kh - 1) * rate + 1, stride)
        pl, pr = _same_pads(xt.shape[3], (kw - 1) * rate + 1, stride)
        xt = F.pad(xt, (pl, pr, pt, pb))
    y = F.conv2d(xt, w_oihw, None, stride=stride, dilation=rate, groups=groups)
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def _finish(y, scope_name, normalizer_fn, normalizer_params, activation_fn, biases: bool, cout: int, collection=None):
    if normalizer_fn is not None:
        y = normalizer_fn(T(y), **(normalizer_params or {}))
        y = _np(y)
    elif biases:
        y = y + _np(get_variable("biases", (cout,)))
    if activation_fn is not None:
        y = _np(activation_fn(T(y)))
    out = T(y.astype(np.float32))
    if collection is not None:                   # slim layers register their output under their scope name
        END_POINTS[scope_name] = out
    return out


@add_arg_scope
def batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, is_training=True, scope=None, **_):
    c = inputs.shape[-1]
    with variable_scope(scope, "BatchNorm"):
        beta = _np(get_variable("beta", (c,)))
        gamma = _np(get_variable("gamma", (c,))) if scale else np.ones(c, np.float32)
        mean, var = _np(get_variable("moving_mean", (c,))), _np(get_variable("moving_variance", (c,)))
    assert not is_training
    xt = _t(inputs).permute(0, 3, 1, 2)
    y = F.batch_norm(xt, torch.from_numpy(mean), torch.from_numpy(var), torch.from_numpy(gamma), torch.from_numpy(beta), False,
                     0.0, epsilon)
    return T(y.permute(0, 2, 3, 1).contiguous().numpy())


@add_arg_scope
def conv2d(inputs, num_outputs, kernel_size, stride=1, padding="SAME", rate=1, activation_fn=None, normalizer_fn=None,
           normalizer_params=None, weights_initializer=None, weights_regularizer=None, biases_initializer=True, reuse=None,
           scope=None, outputs_collections=None, **_):
    k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
    cin = inputs.shape[-1]
    with variable_scope(scope, "Conv") as sc:
        w = _np(get_variable("weights", (k, k, cin, num_outputs)))                       # HWIO
        y = _conv_nhwc(inputs, torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1))), stride, rate, padding)
        LAYERS.append({"op": "conv2d", "scope": sc.name, "k": k, "stride": stride, "rate": rate, "padding": padding,
                       "cin": int(cin), "cout": int(num_outputs), "in_hw": list(inputs.shape[1:3]), "out_hw": list(y.shape[1:3]),
                       "bn": normalizer_fn is not None, "relu": activation_fn is not None})
        return _finish(y, sc.name, normalizer_fn, normalizer_params, activation_fn, biases_initializer is not None, num_outputs,
                       outputs_collections)


@add_arg_scope
def separable_conv2d(inputs, num_outputs, kernel_size, depth_multiplier=1, stride=1, padding="SAME", rate=1,
                     activation_fn=None, normalizer_fn=None, normalizer_params=None, weights_initializer=None,
                     weights_regularizer=None, biases_initializer=True, reuse=None, scope=None, outputs_collections=None, **_):
    assert num_outputs is None and depth_multiplier == 1, "the reference only builds depthwise-only separable convs"
    k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
    c = inputs.shape[-1]
    with variable_scope(scope, "SeparableConv2d") as sc:
        w = _np(get_variable("depthwise_weights", (k, k, c, 1)))                         # [kh, kw, C, 1]
        y = _conv_nhwc(inputs, torch.from_numpy(np.ascontiguousarray(w.transpose(2, 3, 0, 1))), stride, rate, padding, groups=c)
        LAYERS.append({"op": "depthwise", "scope": sc.name, "k": k, "stride": stride, "rate": rate, "padding": padding,
                       "cin": int(c), "cout": int(c), "in_hw": list(inputs.shape[1:3]), "out_hw": list(y.shape[1:3]),
                       "bn": normalizer_fn is not None, "relu": activation_fn is not None})
        return _finish(y, sc.name, normalizer_fn, normalizer_params, activation_fn, biases_initializer is not None, c,
                       outputs_collections)


def resize_bilinear(images, size, align_corners=False, name=None):
    """tf.image.resize_bilinear (TF1): align_corners -> src = dst * (in - 1) / (out - 1); else legacy src = dst * in / out."""
    a = _np(images).astype(np.float32)
    oh, ow = (int(v) for v in (size.a if isinstance(size, T) else [int(_np(s)) for s in size]))
    n, h, w, c = a.shape

    def idx(out, inn):
        scale = np.float32(inn - 1) / np.float32(out - 1) if (align_corners and out > 1) else np.float32(inn) / np.float32(out)
        f = np.arange(out, dtype=np.float32) * scale
        lo = np.floor(f).astype(np.int64)
        hi = np.minimum(lo + 1, inn - 1)
        return lo, hi, (f - lo.astype(np.float32)).astype(np.float32)
    y0, y1, fy = idx(oh, h)
    x0, x1, fx = idx(ow, w)
    top = a[:, y0][:, :, x0] + (a[:, y0][:, :, x1] - a[:, y0][:, :, x0]) * fx[None, None, :, None]
    bot = a[:, y1][:, :, x0] + (a[:, y1][:, :, x1] - a[:, y1][:, :, x0]) * fx[None, None, :, None]
    return T((top + (bot - top) * fy[None, :, None, None]).astype(np.float32))


def install():
    """tfshim.install() + tf.contrib.slim / tf.image.resize_bilinear / resnet_utils; returns the tensorflow stand-in."""
    tf = tfshim.install(is_training=False)
    slim = types.ModuleType("tensorflow.contrib.slim")
    slim.arg_scope, slim.add_arg_scope = arg_scope, add_arg_scope
    slim.conv2d, slim.separable_conv2d, slim.batch_norm = conv2d, separable_conv2d, batch_norm
    slim.l2_regularizer = lambda *a, **k: None
    slim.dropout = add_arg_scope(lambda x, keep_prob=0.5, is_training=True, scope=None, **k: x)
    slim.softmax = lambda x, scope=None: tf.nn.softmax(x)
    def collect_named_outputs(coll, name, out):
        if coll is not None:
            END_POINTS[name] = out
        return out
    slim.utils = types.SimpleNamespace(collect_named_outputs=collect_named_outputs,
                                       convert_collection_to_dict=lambda coll, clear_collection=False: dict(END_POINTS))
    contrib = types.ModuleType("tensorflow.contrib")
    contrib.slim = slim
    nets = types.ModuleType("tensorflow.contrib.slim.nets")
    resnet_utils = types.ModuleType("tensorflow.contrib.slim.nets.resnet_utils")

    def conv2d_same(inputs, num_outputs, kernel_size, stride, rate=1, scope=None):
        """slim resnet_utils.conv2d_same: stride 1 -> SAME; else explicit symmetric-ish pad (beg = total // 2) + VALID."""
        if stride == 1:
            return conv2d(inputs, num_outputs, kernel_size, stride=1, rate=rate, padding="SAME", scope=scope)
        k_eff = kernel_size + (kernel_size - 1) * (rate - 1)
        beg = (k_eff - 1) // 2
        end = k_eff - 1 - beg
        x = tf.pad(inputs, [[0, 0], [beg, end], [beg, end], [0, 0]])
        return conv2d(x, num_outputs, kernel_size, stride=stride, rate=rate, padding="VALID", scope=scope)
    resnet_utils.conv2d_same = conv2d_same
    resnet_utils.subsample = lambda x, factor, scope=None: x if factor == 1 else T(_np(x)[:, ::factor, ::factor])
    nets.resnet_utils = resnet_utils
    slim.nets = nets
    tf.contrib = contrib
    tf.image.resize_bilinear = resize_bilinear
    tf.newaxis = None
    tf.truncated_normal_initializer = lambda *a, **k: None
    tf.logging = types.SimpleNamespace(warning=lambda *a, **k: None, info=lambda *a, **k: None)
    tf.get_variable_scope = lambda: types.SimpleNamespace(name="/".join(s for s in tfshim._SCOPE if s))
    tf.reduce_max = lambda x, axis=None, name=None: T(_np(x).max(axis=axis))
    tf.add_n = lambda xs, name=None: T(sum(_np(x) for x in xs))
    import sys
    sys.modules.update({"tensorflow.contrib": contrib, "tensorflow.contrib.slim": slim, "tensorflow.contrib.slim.nets": nets,
                        "tensorflow.contrib.slim.nets.resnet_utils": resnet_utils})
    return tf


def install_output_layer_api(tf):
    """The extra TF names refinement_net/network/{Layer, Util, SegmentationOutputLayers}.py and core/Measures.py touch at import
    or in SegmentationSoftmax's eval branch (SegmentationOutputLayers.py:17-135): legacy resize_images / nearest-neighbour
    resize, 3-argument where, sparse softmax cross entropy (the loss is built even at inference), py_func, unstack."""
    import sys
    tf.__path__ = []
    for n in ("tensorflow.python", "tensorflow.python.training", "tensorflow.python.training.moving_averages",
              "tensorflow.python.layers", "tensorflow.python.layers.utils"):
        sys.modules[n] = tfshim._Anything(n)
        sys.modules[n].__path__ = []

    def resize_images(images, size, method=0, align_corners=False):
        """tf.image.resize_images (bilinear): 3-D input is one image (expanded and squeezed again)."""
        if len(_np(images).shape) == 3:
            return T(_np(resize_bilinear(T(_np(images)[None]), size, align_corners))[0])
        return resize_bilinear(images, size, align_corners)
    tf.image.resize_images = resize_images

    def resize_nearest_neighbor(images, size, align_corners=False, name=None):
        """TF1 legacy: src = min(floor(dst * in / out), in - 1), the scale and the product in float32."""
        a = _np(images)
        oh, ow = (int(v) for v in (size.a if isinstance(size, T) else [int(_np(s)) for s in size]))
        h, w = a.shape[1:3]
        yi = np.minimum(np.floor(np.arange(oh, dtype=np.float32) * (np.float32(h) / np.float32(oh))).astype(np.int64), h - 1)
        xi = np.minimum(np.floor(np.arange(ow, dtype=np.float32) * (np.float32(w) / np.float32(ow))).astype(np.int64), w - 1)
        return T(a[:, yi][:, :, xi])
    tf.image.resize_nearest_neighbor = resize_nearest_neighbor
    base_softmax, base_where = tf.nn.softmax, tf.where
    tf.nn.softmax = lambda x, axis=-1, name=None: base_softmax(x, name if isinstance(name, str) else None)
    tf.nn.elu = lambda x, name=None: T(np.where(_np(x) > 0, _np(x), np.expm1(_np(x))))
    tf.not_equal = lambda a, b, name=None: T(np.not_equal(_np(a), _np(b)))
    tf.where = lambda c, x=None, y=None, name=None: base_where(c) if x is None else T(np.where(_np(c), _np(x), _np(y)))

    def sparse_softmax_cross_entropy_with_logits(logits=None, labels=None, name=None):
        a = _np(logits).astype(np.float32)
        m = a.max(-1, keepdims=True)
        lse = (m + np.log(np.exp(a - m).sum(-1, keepdims=True)))[..., 0]
        return T((lse - np.take_along_axis(a, _np(labels)[..., None].astype(np.int64), -1)[..., 0]).astype(np.float32))
    tf.nn.sparse_softmax_cross_entropy_with_logits = sparse_softmax_cross_entropy_with_logits
    tf.reduce_sum = lambda x, axis=None, name=None: T(_np(x).sum(axis=tuple(axis) if isinstance(axis, list) else axis))
    tf.unstack = lambda x, axis=0: [T(v) for v in np.moveaxis(_np(x), axis, 0)]

    def py_func(f, inp, Tout, name=None):
        res = [T(np.asarray(o)) for o in f(*[_np(i) for i in inp])]
        for r in res:
            r.set_shape = lambda shape: None
        return res
    tf.py_func = py_func
    if not hasattr(np, "cast"):                       # numpy < 2 spelling used by core/Measures.py:66
        np.cast = type("_Cast", (), {"__getitem__": lambda self, d: (lambda x: np.asarray(x, dtype=d))})()


END_POINTS: Dict[str, T] = {}
