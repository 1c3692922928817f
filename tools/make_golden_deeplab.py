"""Generates tests/golden/deeplab_ref.npz / deeplab_host_refs.json by IMPORTING AND EXECUTING the reference's DeepLabv3+ /
Xception-65 graph code (refinement_net/network/deeplab/{model.py, common.py, core/xception.py, core/feature_extractor.py},
unmodified, with the ModelOptions of DeepLabV3Plus.py:18-32) on tools/slimshim.py -- an eager stand-in for TF 1.8 + slim,
which are absent from the image (see its header for what that does and does not pin) -- plus the pure-python pieces:
datasets/util/BoundingBox.py:15-19, optical_flow_net-PWC-Net/script_pwc_multi.py:16-31 (writeFlowFile, ast-extracted: the
script runs the whole flow stage at import) and MergeTrack/merge_functions.py:197-207 (get_flow, the consumer's reader).

Fixtures (data only; weights / inputs are regenerated from premvos_amd.synth seeds by the tests):
  deeplab_ref.npz           one pass of model.multi_scale_logits on a [1,385,385,4] input, 2 middle-flow units (sub-sampled): end points
                            (entry_flow/block2 skip), Xception output, ASPP output, decoder features, logits
  deeplab_ref_output.npz    SegmentationSoftmax's eval branch (SegmentationOutputLayers.py:17-135, configs/run:34-36) on five crop boxes:
                            logits in, frame-size mask (bit-packed) and foreground posterior out
  deeplab_host_refs.json    the Xception-65 block table xception_65() builds, every conv / depthwise layer the graph code
                            instantiates (scope, kernel, stride, rate, padding, shapes), every variable it requests,
                            scale_dimension values, encode_bbox_as_mask_np cases, .flo bytes

Usage: python tools/make_golden_deeplab.py [/root/reference]"""
import ast
import base64
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True

import slimshim  # noqa: E402
import tfshim  # noqa: E402

tf = slimshim.install()
T = tfshim.T
CODE = os.path.join(REF, "code")
# the deeplab modules import each other relatively (from ..deeplab.core import feature_extractor): give them their package
# path without running refinement_net/network/__init__ machinery (Layer.py etc. need the rest of the TF API)
for pkg, path in (("refinement_net", "refinement_net"), ("refinement_net.network", "refinement_net/network"),
                  ("refinement_net.network.deeplab", "refinement_net/network/deeplab"),
                  ("refinement_net.network.deeplab.core", "refinement_net/network/deeplab/core")):
    m = types.ModuleType(pkg)
    m.__path__ = [os.path.join(CODE, path)]
    sys.modules[pkg] = m
from refinement_net.network.deeplab import common, model  # noqa: E402
from refinement_net.network.deeplab.core import feature_extractor, xception  # noqa: E402

NUM_MIDDLE, SIZE = 2, 385


def block_table():
    """xception_65() builds its list of Blocks and hands it to xception(): capture that list."""
    seen = {}
    real = xception.xception
    xception.xception = lambda inputs, blocks, **kw: seen.update(blocks=blocks, kw=kw) or (inputs, {})
    try:
        xception.xception_65(T(np.zeros((1, 8, 8, 4), np.float32)), output_stride=16)
    finally:
        xception.xception = real
    return [{"scope": b.scope, "num_units": len(b.args), **{k: v for k, v in b.args[0].items() if k != "regularize_depthwise"}}
            for b in seen["blocks"]]


def run_graph():
    from premvos_amd import synth
    from premvos_amd import weights as W
    w = synth.refinement_weights(7, NUM_MIDDLE)
    tfshim.VARIABLES.clear()
    tfshim.VARIABLES.update({k: np.asarray(v) for k, v in W.refinement_weights_to_tf(w).items()})
    tfshim.REQUESTED.clear()
    slimshim.LAYERS.clear()
    slimshim.END_POINTS.clear()
    # a shallower middle flow for the fixture: patch the unit count the reference's own xception_block receives
    real_block = xception.xception_block

    def block(scope, *a, **k):
        if scope.startswith("middle_flow"):
            k["num_units"] = NUM_MIDDLE
        return real_block(scope, *a, **k)
    xception.xception_block = block
    opts = common.ModelOptions(outputs_to_num_classes={"features": 2}, crop_size=None, atrous_rates=[6, 12, 18], output_stride=16,
                               merge_method="max", add_image_level_feature=True, aspp_with_batch_norm=True,
                               aspp_with_separable_conv=True, multi_grid=None, decoder_output_stride=4,
                               decoder_use_separable_conv=True, logits_kernel_size=1, model_variant="xception_65")
    rng = np.random.default_rng(17)
    x01 = rng.random((1, SIZE, SIZE, 4), dtype=np.float32)
    x01[..., 3] = (x01[..., 3] > 0.5)                                   # guidance channel in {0, 1}
    images = T((x01 * 255).astype(np.float32))                           # DeepLabV3Plus.py:12-14: unnormalize(inputs) * 255
    cap = {}
    real_extract, real_refine = model._extract_features, model.refine_by_decoder
    model._extract_features = lambda *a, **k: cap.setdefault("aspp", real_extract(*a, **k))
    model.refine_by_decoder = lambda *a, **k: cap.setdefault("decoder", real_refine(*a, **k))
    try:
        out = model.multi_scale_logits(images, model_options=opts, image_pyramid=None, weight_decay=1e-4, is_training=False,
                                       fine_tune_batch_norm=False)
    finally:
        xception.xception_block = real_block
        model._extract_features, model.refine_by_decoder = real_extract, real_refine
    logits = out["features"]["merged_logits"]
    skip = slimshim.END_POINTS["xception_65/entry_flow/block2/unit_1/xception_module/separable_conv2_pointwise"]
    xc = [v for k, v in slimshim.END_POINTS.items() if k.endswith("exit_flow/block2/unit_1/xception_module/separable_conv3_pointwise")][0]
    # big tensors are stored sub-sampled (the input is regenerated by the tests from its seed)
    arrays = {"skip_sub": skip.a[:, ::4, ::4, ::8], "xception_out_sub": xc.a[:, :, :, ::16], "aspp_sub": cap["aspp"][0].a[:, :, :, ::2],
              "decoder_sub": cap["decoder"].a[:, ::4, ::4, ::8], "logits": logits.a,
              "input_checksum": np.array([float(images.a.astype(np.float64).sum()), float(np.abs(images.a).max())])}
    return arrays, list(slimshim.LAYERS), list(tfshim.REQUESTED)


def output_layer_cases():
    """SegmentationSoftmax (network/SegmentationOutputLayers.py:17-135) constructed as configs/run:34-36 asks (resize_logits) in
    inference mode with batch 1 and a crop box: the eval branch resizes the logits to the label size, soft-maxes, arg-maxes the
    LOGITS, resizes mask (nearest) / foreground probability (bilinear) to the crop size and zero-pads to the frame.  loss="ce"
    instead of the config's "bootstrapped_ce": the loss is built but never fetched at inference."""
    slimshim.install_output_layer_api(tf)
    from refinement_net.core import Extractions
    from refinement_net.datasets import DataKeys
    from refinement_net.network import SegmentationOutputLayers as S
    rng = np.random.default_rng(29)
    H, W = 120, 200
    out = {"frame_hw": np.array([H, W], np.int32)}
    crops = [(0, 0, 120, 200), (10, 20, 100, 190), (50, 60, 52, 63), (0, 150, 120, 200), (37, 3, 119, 61)]
    out["crops"] = np.array(crops, np.int32)
    for i, crop in enumerate(crops):
        coarse = rng.standard_normal((1, 9, 9, 2)).astype(np.float32) * 3
        logits = np.kron(coarse, np.ones((1, 11, 11, 1), np.float32))[:, 1:98, 1:98] + rng.standard_normal((1, 97, 97, 2)).astype(np.float32) * 0.25
        nid = {DataKeys.SEGMENTATION_LABELS: T(np.zeros((1, SIZE, SIZE, 1), np.uint8)),
               DataKeys.SEGMENTATION_LABELS_ORIGINAL_SIZE: T(np.zeros((1, H, W, 1), np.uint8)),
               DataKeys.CROP_BOXES_y0x0y1x1: T(np.array([crop], np.int32))}
        layer = S.SegmentationSoftmax("output", [T(logits)], types.SimpleNamespace(num_classes=lambda: 2), nid,
                                      types.SimpleNamespace(is_training=False, network_name="net"), resize_logits=True, loss="ce")
        mask = layer.extractions[Extractions.SEGMENTATION_MASK_ORIGINAL_SIZE].a
        post = layer.extractions[Extractions.SEGMENTATION_POSTERIORS_ORIGINAL_SIZE].a
        assert mask.shape == post.shape == (1, H, W) and mask.dtype == np.int64 and post.dtype == np.float32
        out[f"logits{i}"] = logits[0].astype(np.float32)
        out[f"mask{i}"] = np.packbits(mask[0].astype(np.uint8))
        out[f"post{i}"] = post[0]
        out[f"post385_{i}"] = layer.extractions[Extractions.SEGMENTATION_POSTERIORS].a[0, ::4, ::4]
    return out, sorted(layer.extractions)


def extract_function(path, name):
    src = open(path).read()
    node = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    return ast.get_source_segment(src, node)


def main():
    os.makedirs(GOLD, exist_ok=True)
    g = {"num_middle": NUM_MIDDLE, "size": SIZE, "weights": "premvos_amd.synth.refinement_weights(7, num_middle)",
         "input": "x = default_rng(17).random((1, size, size, 4), float32); x[..., 3] = x[..., 3] > 0.5; images = (x * 255).astype(float32)"}
    g["blocks"] = block_table()
    arrays, layers, requested = run_graph()
    g["layers"] = layers
    g["variables"] = [[n, list(s)] for n, s in requested]
    g["scale_dimension"] = [[d, s, model.scale_dimension(d, s)] for d in (385, 129, 513, 97, 25, 7, 1) for s in (0.25, 1.0 / 16, 0.5, 1.0, 2.0)]
    g["decoder_end_point"] = feature_extractor.networks_to_feature_maps["xception_65"][feature_extractor.DECODER_END_POINTS]
    # BoundingBox.py imports tensorflow at module level: the stand-in is enough for the numpy function we want
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_bbox", os.path.join(CODE, "refinement_net", "datasets", "util", "BoundingBox.py"))
    bb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bb)
    cases = []
    for box in ([2.4, 3.5, 7.5, 9.49], [0.5, 1.5, 2.5, 3.5], [-3.2, -1.0, 4.6, 5.0], [10.0, 2.0, 30.0, 40.0], [3.0, 3.0, 3.4, 3.4]):
        m = bb.encode_bbox_as_mask_np(np.array(box, np.float32), (12, 14, 3))
        cases.append({"bbox_y0x0y1x1": box, "shape": [12, 14], "mask": m[:, :, 0].tolist()})
    g["encode_bbox_as_mask_np"] = cases
    # .flo writer (flow stage) and the reader MergeTrack uses on it
    ns = {"np": np, "sys": sys}
    exec(extract_function(os.path.join(CODE, "optical_flow_net-PWC-Net", "script_pwc_multi.py"), "writeFlowFile"), ns)
    exec(extract_function(os.path.join(CODE, "MergeTrack", "merge_functions.py"), "get_flow"), ns)
    uv = np.random.default_rng(4).standard_normal((5, 7, 2)).astype(np.float32)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fn = os.path.join(td, "x.flo")
        ns["writeFlowFile"](fn, uv)
        raw = open(fn, "rb").read()
        back = ns["get_flow"](fn)
    g["flo"] = {"uv": uv.tolist(), "bytes_b64": base64.b64encode(raw).decode(), "reader_roundtrip_equal": bool(np.array_equal(back, uv))}
    np.savez_compressed(os.path.join(GOLD, "deeplab_ref.npz"), **arrays)
    ol, g["output_layer_extractions"] = output_layer_cases()
    np.savez_compressed(os.path.join(GOLD, "deeplab_ref_output.npz"), **ol)
    with open(os.path.join(GOLD, "deeplab_host_refs.json"), "w") as f:
        json.dump(g, f, indent=1)
    for fn in ("deeplab_ref.npz", "deeplab_ref_output.npz", "deeplab_host_refs.json"):
        print(fn, os.path.getsize(os.path.join(GOLD, fn)), "bytes")
    print("layers:", len(layers), "variables:", len(requested), "logits", arrays["logits"].shape)


if __name__ == "__main__":
    main()
