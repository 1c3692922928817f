"""Generates tests/golden/reid_ref.npz / reid_host_refs.json by IMPORTING AND EXECUTING the reference's ReID_net python in the
build container (where /root/reference exists), unmodified, on tools/tfshim.py (+ the extra TF names below), an eager
numpy/torch stand-in for TensorFlow 1.x, which is absent from the image:

  * ReID_net/Config.py reads ReID_net/configs/run; network/Network.py:build_tower walks its "network" table and instantiates
    network/NetworkLayers.py (Conv, ResidualUnit2, FullyConnected) and network/NetworkOutputLayers.py
    (FullyConnectedWithTripletLoss) exactly as Engine does at inference (is_training False, freeze_batchnorm True);
  * datasets/Similarity/DAVIS_Forward_Feed.py builds the crops the in-merge ReID service feeds (context region 1.2, tf.round,
    clip with an excess of at least one pixel, zeros for boxes with min(h, w) <= 10, resize_images to 128x128, normalize).

What this pins: the layer table of configs/run, every layer's wiring / kernel / stride / BatchNorm placement, the checkpoint
variable names and shapes, the float32 box arithmetic and crop / resize / normalise order.  What it cannot pin: conv2d /
max_pool 'SAME', batch_normalization, resize_images and round themselves are restated in the stand-in from the published TF
semantics (third-party, absent).  The triplet loss FullyConnectedWithTripletLoss builds next to the embedding is never fetched
at inference: tf.map_fn over it is a no-op here.

Fixtures are data only (weights / frames are regenerated from seeds by the tests).
Usage: python tools/make_golden_reid.py [/root/reference]"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True

import slimshim  # noqa: E402
import tfshim  # noqa: E402
from tfshim import T, _np  # noqa: E402

tf = slimshim.install()
slimshim.install_output_layer_api(tf)
tfshim.STUB_ROOTS += ("skimage", "partialflow")
os.environ.setdefault("USER", "nobody")       # datasets/Util/Util.py:55 builds a default data path from it
CODE = os.path.join(REF, "code")
pkg = types.ModuleType("ReID_net")
pkg.__path__ = [os.path.join(CODE, "ReID_net")]
sys.modules["ReID_net"] = pkg

FEED = []                    # values the next tf.placeholder calls take (eager: a placeholder needs its value when it is made)


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(_np(x), dtype=np.float32))


def install_reid_api():
    """TF names network/NetworkLayers.py, NetworkOutputLayers.py:253-272, Util_Network.py and DAVIS_Forward_Feed.py touch."""
    def conv2d(x, W, strides, padding="SAME", name=None):
        assert padding == "SAME" and strides[0] == strides[3] == 1
        xt = _t(x).permute(0, 3, 1, 2)
        wt = _t(W).permute(3, 2, 0, 1).contiguous()
        pt, pb = slimshim._same_pads(xt.shape[2], wt.shape[2], strides[1])
        pl, pr = slimshim._same_pads(xt.shape[3], wt.shape[3], strides[2])
        y = F.conv2d(F.pad(xt, (pl, pr, pt, pb)), wt, stride=(strides[1], strides[2]))
        return T(y.permute(0, 2, 3, 1).contiguous().numpy())
    tf.nn.conv2d = conv2d

    def max_pool(x, ksize, strides, padding="SAME", name=None):
        assert padding == "SAME"
        xt = _t(x).permute(0, 3, 1, 2)
        pt, pb = slimshim._same_pads(xt.shape[2], ksize[1], strides[1])
        pl, pr = slimshim._same_pads(xt.shape[3], ksize[2], strides[2])
        y = F.max_pool2d(F.pad(xt, (pl, pr, pt, pb), value=float("-inf")), (ksize[1], ksize[2]), (strides[1], strides[2]))
        return T(y.permute(0, 2, 3, 1).contiguous().numpy())
    tf.nn.max_pool = max_pool

    def batch_normalization(x, mean, variance, offset, scale, variance_epsilon, name=None):
        """tf.nn.batch_normalization: inv = rsqrt(var + eps) * scale; x * inv + (offset - mean * inv), all float32."""
        inv = (np.float32(1) / np.sqrt(_np(variance) + np.float32(variance_epsilon))).astype(np.float32) * _np(scale)
        return T((_np(x) * inv + (_np(offset) - _np(mean) * inv)).astype(np.float32))
    tf.nn.batch_normalization = batch_normalization
    tf.nn.l2_loss = lambda w, name=None: T(np.float32(0))
    tf.nn.dropout = lambda x, keep_prob, **k: x
    tf.nn.softplus = lambda x, name=None: T(np.logaddexp(_np(x), 0).astype(np.float32))
    tf.get_variable = lambda name, shape=None, dtype=None, initializer=None, trainable=True, **k: tfshim.get_variable(name, tuple(shape))
    tf.constant_initializer = lambda *a, **k: None
    tf.contrib.layers = types.SimpleNamespace(variance_scaling_initializer=lambda *a, **k: None)

    class _Ctx:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False
    tf.device = lambda *a, **k: _Ctx()
    tf.control_dependencies = lambda *a, **k: _Ctx()
    tf.matmul = lambda a, b, name=None: T((_t(a) @ _t(b)).numpy())
    tf.norm = lambda x, axis=None, name=None: T(np.sqrt((_np(x).astype(np.float32) ** 2).sum(axis=axis)).astype(np.float32))
    tf.summary.histogram = lambda *a, **k: None
    # the loss sub-graph is never fetched at inference: not executed (zeros of the declared dtypes)
    tf.map_fn_real = tf.map_fn
    tf.round = lambda x, name=None: T(np.rint(_np(x)))                                 # half to even, like TF
    tf.fill = lambda dims, value, name=None: T(np.full(tfshim._ints(dims), value))
    tf.placeholder = lambda dtype, shape=None, name=None: T(np.asarray(FEED.pop(0), dtype=dtype.np))
    tf.GraphKeys = types.SimpleNamespace(UPDATE_OPS="update_ops")


install_reid_api()

from ReID_net.Config import Config  # noqa: E402
from ReID_net.network import Network  # noqa: E402
from ReID_net.network.Util_Network import TowerSetup  # noqa: E402

N_CROPS, SEED = 3, 5


def run_network(cfg):
    """Network.build_tower on the config's own network table; weights = oracle.reid_oracle.synth_weights(SEED) in TF layout."""
    from oracle import reid_oracle as R
    from premvos_amd import weights as W
    w = R.synth_weights(SEED)
    tfshim.VARIABLES.clear()
    tfshim.VARIABLES.update({k: np.asarray(v) for k, v in W.reid_weights_to_tf(w).items()})
    tfshim.REQUESTED.clear()
    x = np.random.default_rng(SEED).standard_normal((N_CROPS, 128, 128, 3)).astype(np.float32)
    net = object.__new__(Network.Network)
    net.config, net.use_partialflow, net.summaries = cfg, False, []
    net.inputs_tensors_dict = {"original_labels": T(np.ones(N_CROPS, np.int32))}
    tower = TowerSetup(dtype=tf.float32, gpu=0, is_main_train_tower=False, is_training=False, freeze_batchnorm=True,
                       variable_device="/gpu:0", use_update_ops_collection=False, batch_size=N_CROPS, original_sizes=None,
                       resized_sizes=None, use_weight_summaries=False)
    lazy = lambda fn, elems, dtype=None, **k: tuple(T(np.zeros((len(_np(elems)),), d.np)) for d in dtype)   # noqa: E731
    tf.map_fn = lazy
    try:
        out = net.build_tower(cfg.dict("network"), T(x), T(np.ones(N_CROPS, np.int32)), None, 255, cfg.int("num_classes"), tower)
    finally:
        tf.map_fn = tf.map_fn_real
    layers = out[-1]
    arrays = {"input_checksum": np.array([float(x.astype(np.float64).sum())]), "embedding": np.asarray(out[2].a)}
    shapes = {}
    for name, layer in layers.items():
        a = np.asarray(layer.outputs[0].a)
        shapes[name] = list(a.shape)
        if a.ndim == 4:                                             # sub-sampled: every 3rd pixel, every 32nd channel
            arrays["act_" + name] = a[:, ::3, ::3, ::32]
        else:
            arrays["act_" + name] = a
    return arrays, shapes, list(tfshim.REQUESTED)


def run_crops(cfg):
    """DAVISForwardFeedDataset._create_inputs_for_eval on a seeded frame and boxes (xywh), incl. a box leaving the frame and a
    small one (min(h, w) <= 10 after the context region -> zeros)."""
    from ReID_net.datasets.Similarity.DAVIS_Forward_Feed import DAVISForwardFeedDataset
    rng = np.random.default_rng(9)
    h, w = 96, 150
    frame = (rng.random((h, w, 3)) * 255).astype(np.uint8).astype(np.float32)
    boxes = np.array([[10.2, 12.7, 60.5, 40.1], [100.0, 50.0, 70.0, 60.0], [-4.0, -3.0, 30.0, 25.5], [40.0, 30.0, 9.0, 30.0],
                      [0.0, 0.0, 150.0, 96.0], [20.5, 10.5, 33.5, 47.5]], np.float32)
    FEED[:] = [frame, boxes.copy()]
    ds = DAVISForwardFeedDataset(cfg, "valid", None)
    ctx = ds.apply_contex_region(T(boxes.copy()), tf.shape(ds.image))
    FEED[:] = [frame, boxes.copy()]
    ds = DAVISForwardFeedDataset(cfg, "valid", None)
    imgs = ds._create_inputs_for_eval(len(boxes))
    return {"crop_frame": frame.astype(np.uint8), "crop_boxes_xywh": boxes, "crop_context_boxes": np.asarray(ctx.a),
            "crops_sub": np.asarray(imgs.a)[:, ::3, ::3],
            "crops_mean": np.asarray(imgs.a).mean(axis=(1, 2), dtype=np.float64).astype(np.float32)}


def run_similarity_crops(cfg):
    """SimilarityDataset._load_crop_helper (datasets/Similarity/Similarity.py:264-298), the crop of the BATCH stage
    (DAVIS_Forward_Similarity): image via load_image_tensorflow = decode + tf.image.convert_image_dtype(float32), i.e. uint8 *
    float32(1/255) -- NOT the /255 of the in-merge feed --, context region, tf.round, excess >= 0, slice, resize_images, normalize.
    The decoder is replaced by the frame itself; convert_image_dtype is restated (cast, then multiply by 1/max)."""
    from ReID_net.datasets.Similarity import Similarity as SM
    rng = np.random.default_rng(13)
    h, w = 90, 140
    frame = (rng.random((h, w, 3)) * 255).astype(np.uint8)
    boxes = np.array([[10.2, 12.7, 60.5, 40.1], [95.0, 50.0, 70.0, 60.0], [-4.0, -3.0, 30.0, 25.5], [40.0, 30.0, 9.0, 30.0],
                      [0.0, 0.0, 140.0, 90.0], [20.5, 10.5, 33.5, 47.5]], np.float32)
    SM.load_image_tensorflow = lambda fn, jpg, channels=None: T(frame.astype(np.float32) * np.float32(1.0 / 255.0))
    me = types.SimpleNamespace(jpg=True, context_region_factor=cfg.float("context_region_factor_val", 1.2),
                               input_size=tuple(cfg.int_list("input_size")), augmentors=[])
    crops, raw_shapes = [], []
    for b in boxes:
        norm, img, cropped = SM.SimilarityDataset._load_crop_helper(me, "frame.jpg", T(b.copy()))
        crops.append(np.asarray(norm.a))
        raw_shapes.append(list(np.asarray(cropped.a).shape[:2]))
    crops = np.stack(crops)
    return {"sim_frame": frame, "sim_boxes_xywh": boxes, "sim_crop_hw": np.array(raw_shapes, np.int32), "sim_crops_sub": crops[:, ::3, ::3],
            "sim_crops_mean": crops.mean(axis=(1, 2), dtype=np.float64).astype(np.float32)}


def main():
    os.makedirs(GOLD, exist_ok=True)
    cfg = Config(os.path.join(CODE, "ReID_net", "configs", "run"))
    cfg.initialize()
    arrays, shapes, requested = run_network(cfg)
    arrays.update(run_crops(cfg))
    arrays.update(run_similarity_crops(cfg))
    g = {"weights": f"oracle.reid_oracle.synth_weights({SEED})", "input": f"default_rng({SEED}).standard_normal(({N_CROPS},128,128,3), float32)",
         "layer_shapes": shapes, "variables": [[n, list(s)] for n, s in requested],
         "config": {k: cfg._entries[k] for k in ("input_size", "num_classes", "context_region_factor", "output_embedding_layer")},
         "layer_order": list(cfg.dict("network").keys())}
    np.savez_compressed(os.path.join(GOLD, "reid_ref.npz"), **arrays)
    with open(os.path.join(GOLD, "reid_host_refs.json"), "w") as f:
        json.dump(g, f, indent=1)
    for fn in ("reid_ref.npz", "reid_host_refs.json"):
        print(fn, os.path.getsize(os.path.join(GOLD, fn)), "bytes")
    print("embedding", arrays["embedding"].shape, float(np.abs(arrays["embedding"]).max()), "variables", len(requested))


if __name__ == "__main__":
    main()
