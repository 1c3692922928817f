"""CPU tests of the refinement oracle + host logic (no GPU)."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import refinement_oracle as R
from premvos_amd import rle


def test_module_plan_matches_oracle_and_stride_to_atrous_switch():
    from premvos_amd.refinement import module_plan
    a, b = module_plan(16), R.plan_modules(16)
    assert a == b and len(a) == 21
    # entry blocks stride 2; middle stride 1 rate 1; exit block1 stride->1 (rate 1), exit block2 rate 2
    assert [m[5] for m in a[:3]] == [2, 2, 2] and all(m[5:] == (1, 1) for m in a[3:20])
    assert a[19][0].startswith("exit_flow/block1") and a[20][5:] == (1, 2) and a[20][4] is True
    assert R.scale_dimension(385, 0.25) == 97 and R.scale_dimension(385, 1 / 16) == 25


def test_tf_resize_semantics():
    x = torch.arange(5, dtype=torch.float32).view(1, 1, 1, 5).repeat(1, 1, 2, 1)
    # legacy: src = dst * in/out (no half-pixel shift), clamp at the edge
    assert torch.allclose(R.resize_bilinear_tf(x, 2, 10, False)[0, 0, 0],
                          torch.tensor([0, .5, 1, 1.5, 2, 2.5, 3, 3.5, 4, 4.0]))
    # align_corners: end points map to end points
    assert torch.allclose(R.resize_bilinear_tf(x, 2, 9, True)[0, 0, 0], torch.linspace(0, 4, 9))
    assert torch.equal(R.resize_nearest_tf(x, 2, 10)[0, 0, 0], torch.tensor([0., 0, 1, 1, 2, 2, 3, 3, 4, 4]))
    ref = F.interpolate(x, size=(2, 9), mode="bilinear", align_corners=True)
    assert torch.allclose(R.resize_bilinear_tf(x, 2, 9, True), ref, atol=1e-6)


def test_crop_box_and_guidance():
    assert R.crop_box([20.4, 30.5, 90.6, 150.5], 120, 200) == (0, 0, 120, 200)
    assert R.crop_box([100.5, 101.5, 140.0, 150.0], 480, 854) == (50, 52, 190, 200)     # round half to even
    img = np.zeros((300, 400, 3), np.uint8)
    x, crop = R.make_input(img, [100.0, 120.0, 160.0, 200.0])
    assert crop == (50, 70, 210, 250) and x.shape == (1, 4, 385, 385)
    g = x[0, 3]
    assert set(g.unique().tolist()) == {0.0, 1.0}
    assert abs(g.mean().item() - (60 * 80) / (160 * 180)) < 0.01
    z = R.deeplab_preprocess(x)
    assert set(z[0, 3].unique().tolist()) == {-1.0, 1.0} and torch.allclose(z[0, :3], torch.full((3, 385, 385), -1.0), atol=1e-5)


def test_rle_roundtrip_and_format():
    rng = np.random.default_rng(0)
    for shape in ((7, 9), (1, 1), (480, 854), (5, 3)):
        m = (rng.random(shape) > 0.6).astype(np.uint8)
        e = rle.encode(m * 255)            # the forwarder encodes mask*255
        assert e == R.rle_encode(m) and np.array_equal(rle.decode(e), m) and rle.area(e) == int(m.sum())
        ys, xs = np.nonzero(m)
        if len(ys):
            assert rle.to_bbox(e) == [float(xs.min()), float(ys.min()), float(xs.max() - xs.min() + 1),
                                      float(ys.max() - ys.min() + 1)]
        json.dumps(e)
    assert rle.encode(np.array([[0, 1], [1, 1]], np.uint8)) == {"size": [2, 2], "counts": "13"}
    assert rle.counts_from_mask(np.ones((2, 2))).tolist() == [0, 4]
    # long runs need multi-character groups and negative deltas need the sign bit
    m = np.zeros((300, 300), np.uint8)
    m[10:200, 5:7] = 1
    m[0:3, 100] = 1
    assert np.array_equal(rle.decode(rle.encode(m)), m)
    assert rle.to_bbox(rle.encode(np.zeros((4, 5), np.uint8))) == [0.0, 0.0, 0.0, 0.0]


def test_conf_score_definition():
    mask = np.array([[1, 0], [0, 0]], np.uint8)
    post = np.array([[0.9, 0.2], [0.0, 0.4]], np.float32)
    # (2*0.9-1) + (1-2*0.2) + 1 + (1-2*0.4) over 4
    assert abs(float(R.conf_score(mask, post)) - (0.8 + 0.6 + 1.0 + 0.2) / 4) < 1e-6


def test_config_reader(tmp_path):
    from premvos_amd.refinement import Config
    fn = tmp_path / "run"
    fn.write_text('{\n# a comment line\n"image_input_dir":"../data/x",\n"batch_size_eval": 1,\n'
                  '"input_size_train": [385, 385],\n"use_bbox_guidance": true,\n"load": "w.pt"\n}\n')
    c = Config(str(fn))
    assert c.dir("image_input_dir") == "../data/x/" and c.int("batch_size_eval") == 1
    assert c.int_list("input_size_train") == [385, 385] and c.bool("use_bbox_guidance") is True
    assert c.string("missing", "dflt") == "dflt"
    with pytest.raises(AssertionError):          # core/Config.py:27 "assert default is not None"
        c.string("missing")
    # the shipped reference config parses with the same reader semantics (file content not copied: built here)
    c2 = Config(str(fn), '{"load": "other"}')
    assert c2.string("load") == "other"


def test_oracle_small_forward_and_json():
    w = R.synth_weights(0, 1)
    img = (np.random.default_rng(0).random((90, 140, 3)) * 255).astype(np.uint8)
    props = [{"bbox": [10.5, 20.0, 60.0, 40.5], "score": 0.91}]
    out = R.refine_proposals(w, img, props, 1)
    assert out[0]["segmentation"]["size"] == [90, 140] and isinstance(out[0]["conf_score"], str)
    assert -1.0 <= float(out[0]["conf_score"]) <= 1.0
    json.dumps(out)
