"""oracle/refinement_oracle.py and the product's tables against fixtures produced by EXECUTING the reference's DeepLabv3+ /
Xception-65 graph code (tools/make_golden_deeplab.py: network/deeplab/{model,common}.py, core/{xception,feature_extractor}.py
unmodified, with DeepLabV3Plus.py's ModelOptions, on tools/slimshim.py's eager stand-in for TF 1.8 + slim) and the reference's
pure-python pieces (BoundingBox.py, writeFlowFile, MergeTrack's get_flow).  Pins: the block table, every layer's scope /
kernel / stride / rate / padding / shapes, the composition of the whole net, the checkpoint variable names; conv / batch-norm
/ resize primitives themselves are restated in the stand-in (third-party)."""
import base64
import json
import os

import numpy as np
import torch

from oracle import refinement_oracle as RO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HR = json.load(open(os.path.join(GOLD, "deeplab_host_refs.json")))
REF = np.load(os.path.join(GOLD, "deeplab_ref.npz"))


def ref_input():
    x = np.random.default_rng(17).random((1, HR["size"], HR["size"], 4), dtype=np.float32)
    x[..., 3] = (x[..., 3] > 0.5)
    images = (x * 255).astype(np.float32)
    assert abs(float(images.astype(np.float64).sum()) - REF["input_checksum"][0]) < 1e-3
    return images


def test_xception_block_table():
    from premvos_amd.refinement import model as M
    for blocks in (RO.BLOCKS, M.BLOCKS):
        assert len(blocks) == len(HR["blocks"]) == 6
        for (scope, depths, skip, relu_in, units, stride), ref in zip(blocks, HR["blocks"]):
            assert scope == ref["scope"] and list(depths) == ref["depth_list"] and skip == ref["skip_connection_type"]
            assert relu_in == ref["activation_fn_in_separable_conv"] and units == ref["num_units"] and stride == ref["stride"]
            assert ref["unit_rate_list"] == [1, 1, 1]
    assert HR["decoder_end_point"] == [RO.DECODER_SKIP] and M.DECODER_SKIP + "_pointwise" == RO.DECODER_SKIP


def test_layer_geometry_of_the_reference_graph():
    """Every depthwise layer the reference graph code instantiates (scope order, stride, atrous rate, map sizes) is what the
    oracle's and the product's module plans produce: the stride -> atrous switch at output stride 16 (xception.py:330-345),
    SAME vs explicit padding (xception.py:70-89), the ASPP rates and the decoder."""
    from premvos_amd.refinement import model as M
    dw = [l for l in HR["layers"] if l["op"] == "depthwise"]
    body = [l for l in dw if "/xception_module/" in l["scope"]]
    for plan in (RO.plan_modules(HR["num_middle"]), M.module_plan(HR["num_middle"])):
        exp = []
        for prefix, cin, depths, skip, relu_in, stride, rate in plan:
            for i in range(3):
                exp.append((f"xception_65/{prefix}/separable_conv{i + 1}_depthwise", stride if i == 2 else 1, rate))
        got = [(l["scope"], l["stride"], l["rate"]) for l in body]
        assert got == exp
    # stride-2 layers are the explicitly padded VALID ones, everything else SAME; sizes 385 -> 193 -> 97 -> 49 -> 25
    for l in body:
        assert l["padding"] == ("VALID" if l["stride"] == 2 else "SAME")
    assert [l["out_hw"][0] for l in body if l["stride"] == 2] == [97, 49, 25]
    assert all(l["out_hw"] == [25, 25] for l in body if "exit_flow" in l["scope"] or "middle_flow" in l["scope"])
    assert [l["rate"] for l in body if "exit_flow/block2" in l["scope"]] == [2, 2, 2]
    aspp = [l for l in dw if l["scope"].startswith("aspp")]
    assert [(l["scope"], l["rate"]) for l in aspp] == [(f"aspp{i}_depthwise", r) for i, r in enumerate(RO.ATROUS_RATES, 1)]
    dec = [l for l in HR["layers"] if l["scope"].startswith("decoder/")]
    assert [l["scope"] for l in dec] == ["decoder/feature_projection0", "decoder/decoder_conv0_depthwise", "decoder/decoder_conv0_pointwise",
                                         "decoder/decoder_conv1_depthwise", "decoder/decoder_conv1_pointwise"]
    assert dec[0]["cin"] == 256 and dec[0]["cout"] == 48 and dec[1]["cin"] == 304 and dec[1]["in_hw"] == [97, 97]
    stem = HR["layers"][0]
    assert stem["scope"].endswith("entry_flow/conv1_1") and stem["stride"] == 2 and stem["padding"] == "VALID" and stem["in_hw"] == [387, 387]
    assert HR["layers"][-1]["scope"] == "logits/features" and not HR["layers"][-1]["bn"] and not HR["layers"][-1]["relu"]


def test_scale_dimension_and_bbox_guidance_and_flo_format():
    for d, s, v in HR["scale_dimension"]:
        assert RO.scale_dimension(d, s) == v
    for c in HR["encode_bbox_as_mask_np"]:
        b = c["bbox_y0x0y1x1"]
        if min(b) < 0:
            continue          # numpy's negative-index slicing (BoundingBox.py:18) is unreachable: boxes are clipped, eval.py:94
        h, w = c["shape"]
        guid = np.zeros((h, w), np.uint8)
        y0, x0, y1, x1 = (int(v) for v in np.round(np.asarray(b, np.float32)))
        guid[max(y0, 0):max(y1, 0), max(x0, 0):max(x1, 0)] = 1          # oracle/refinement_oracle.py: make_input
        assert guid.tolist() == c["mask"], b
    # .flo: the product writer produces the reference writer's bytes; the product reader == the consumer's reader
    from premvos_amd.flow.driver import readFlowFile, writeFlowFile
    import tempfile
    uv = np.array(HR["flo"]["uv"], np.float32)
    with tempfile.TemporaryDirectory() as td:
        fn = os.path.join(td, "a.flo")
        writeFlowFile(fn, uv)
        assert open(fn, "rb").read() == base64.b64decode(HR["flo"]["bytes_b64"])
        assert np.array_equal(readFlowFile(fn), uv) and HR["flo"]["reader_roundtrip_equal"]


def _close(a, b, tol):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() <= tol * max(1.0, float(np.abs(b).max()))


def test_whole_graph_oracle_vs_reference_multi_scale_logits():
    from premvos_amd import synth
    w = synth.refinement_weights(7, HR["num_middle"])
    images = ref_input()
    # the oracle takes the NORMALISED network input and undoes the normalisation like DeepLabV3Plus.py:12-14
    x01 = torch.from_numpy(images / 255).permute(0, 3, 1, 2)
    mean = torch.cat([torch.from_numpy(RO.IMAGENET_RGB_MEAN), torch.zeros(1)]).view(1, 4, 1, 1)
    std = torch.cat([torch.from_numpy(RO.IMAGENET_RGB_STD), torch.ones(1)]).view(1, 4, 1, 1)
    inter = {}
    with torch.no_grad():
        lg = RO.deeplab_logits(w, (x01 - mean) / std, HR["num_middle"], inter)
    nhwc = lambda t: t.permute(0, 2, 3, 1).numpy()                                   # noqa: E731
    assert _close(nhwc(inter["skip"])[:, ::4, ::4, ::8], REF["skip_sub"], 2e-4)
    assert _close(nhwc(inter["xception"])[:, :, :, ::16], REF["xception_out_sub"], 2e-4)
    assert _close(nhwc(inter["aspp"])[:, :, :, ::2], REF["aspp_sub"], 2e-4)
    assert _close(nhwc(inter["decoder"])[:, ::4, ::4, ::8], REF["decoder_sub"], 2e-4)
    assert _close(nhwc(lg), REF["logits"], 2e-4)


def test_variable_names_the_graph_requests_are_the_importers():
    from premvos_amd import synth
    from premvos_amd import weights as W
    req = {n: tuple(s) for n, s in HR["variables"]}
    w = synth.refinement_weights(7, HR["num_middle"])
    tfv = W.refinement_weights_to_tf(w)
    assert sorted(req) == sorted(tfv), (sorted(set(req) ^ set(tfv))[:6])
    assert all(tuple(tfv[n].shape) == s for n, s in req.items())
    back = W.refinement_weights_from_tf({n: tfv[n] for n in req})
    assert sorted(back) == sorted(w)
    for k in w:
        a, b = w[k], back[k]
        if isinstance(a, dict):
            assert all(torch.equal(a[j], b[j]) for j in a), k
        else:
            assert torch.equal(a, b), k


def test_output_layer_oracle_vs_reference_segmentation_softmax():
    """SegmentationSoftmax's eval branch (network/SegmentationOutputLayers.py:17-135, built as configs/run:34-36 says) executed on
    the stand-in: frame-size mask and foreground posterior for five crop boxes (full frame, interior, 2x3 pixels, borders)."""
    ref = np.load(os.path.join(GOLD, "deeplab_ref_output.npz"))
    assert HR["output_layer_extractions"] == sorted(["segmentation_posteriors", "segmentation_mask_original_size",
                                                      "segmentation_posteriors_original_size"])
    h, w = (int(v) for v in ref["frame_hw"])
    for i, crop in enumerate(ref["crops"]):
        lg = torch.from_numpy(ref[f"logits{i}"]).permute(2, 0, 1)[None].contiguous()
        mask, post = RO.output_layer(lg, tuple(int(v) for v in crop), h, w)
        want = np.unpackbits(ref[f"mask{i}"])[:h * w].reshape(h, w)
        assert np.array_equal(mask, want), i
        assert np.abs(post - ref[f"post{i}"]).max() < 1e-6, i
        # the 385x385 probability the crop-size one is resized from
        p385 = torch.softmax(RO.resize_bilinear_tf(lg, RO.INPUT_SIZE, RO.INPUT_SIZE, False), dim=1)[0, 1].numpy()
        assert np.abs(p385[::4, ::4] - ref[f"post385_{i}"]).max() < 1e-6
