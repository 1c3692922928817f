"""oracle/reid_oracle.py and the host side of premvos_amd/reid (no GPU): TF 'SAME' geometry, the context-region box
arithmetic of both crop variants, weight-name mapping."""
import numpy as np
import torch

from oracle import reid_oracle as R


def test_same_padding_geometry():
    assert R.same_pad(128, 3, 1) == (128, 1, 1)
    assert R.same_pad(128, 3, 2) == (64, 0, 1)            # even size, stride 2: the one pad pixel goes AFTER
    assert R.same_pad(7, 3, 2) == (4, 1, 1)
    assert R.same_pad(4, 3, 3) == (2, 1, 1)               # the 3x3/3 max-pool on the 4x4 map
    assert R.same_pad(8, 1, 2) == (4, 0, 0)
    assert R.final_spatial() == 2
    x = torch.arange(16.0).view(1, 1, 4, 4)
    assert R.max_pool_same(x, 3, 3).flatten().tolist() == [5.0, 7.0, 13.0, 15.0]
    w = torch.ones(1, 1, 3, 3)
    y = R.conv_same(torch.ones(1, 1, 4, 4), w, 2)         # windows start at 0 and 2; the pad column/row is at the END
    assert y.flatten().tolist() == [9.0, 6.0, 6.0, 4.0]


def test_context_boxes_both_variants_and_product_twin():
    from premvos_amd.reid import context_boxes
    boxes = [[10.5, 20.0, 60.0, 40.5], [150.0, 80.0, 80.0, 60.0], [0.0, 0.0, 8.0, 30.0], [2.5, 2.5, 5.0, 5.0],
             [190.0, 110.0, 30.0, 30.0]]
    feed = R.context_boxes(boxes, 120, 200, True)
    batch = R.context_boxes(boxes, 120, 200, False)
    # box 0: x = 10.5 - 6 = 4.5 -> round-half-even 4; y = 20 - 4.05 -> 16; w = 72, h = 48.6 -> 49
    assert batch[0].tolist() == [4, 16, 72, 49] and feed[0].tolist() == [4, 16, 71, 48]      # feed: excess >= 1
    # box 1 runs over the right/bottom edge: w clipped to the image in the batch variant, one pixel more in the feed one
    assert batch[1].tolist() == [142, 74, 58, 46] and feed[1].tolist() == [142, 74, 58, 46]
    assert batch[4][0] + batch[4][2] == 200 and batch[4][1] + batch[4][3] == 120
    assert np.array_equal(context_boxes(boxes, 120, 200, True), feed)
    assert np.array_equal(context_boxes(boxes, 120, 200, False), batch)
    assert feed.dtype == np.int32


def test_small_boxes_give_zero_images_only_in_the_feed_variant():
    img = np.random.default_rng(0).integers(0, 256, (60, 80, 3), dtype=np.uint8)
    c = R.make_crop(img, [3, 4, 10, 40], feed=True)
    assert np.allclose(c, (0 - R.IMAGENET_RGB_MEAN) / R.IMAGENET_RGB_STD)
    c2 = R.make_crop(img, [3, 4, 10, 40], feed=False)
    assert c2.std() > 0.1 and c2.shape == (128, 128, 3)
    # resize: an 1:1 crop is the image itself
    c3 = R.make_crop(np.tile(img[:1, :1], (128, 128, 1)), [0, 0, 128, 128], feed=False)
    assert np.allclose(c3, (img[0, 0].astype(np.float32) / 255 - R.IMAGENET_RGB_MEAN) / R.IMAGENET_RGB_STD, atol=1e-6)


def test_weight_names_round_trip_through_a_tf_checkpoint(tmp_path):
    from premvos_amd import weights as W
    units = [R.UNITS[0], ("res15", 3, (16, 32, 64), (1, 3, 1), (1, 2, 1))]
    w = R.synth_weights(1, units)
    v = W.reid_weights_to_tf(w)
    assert v["res0/W1"].shape == (3, 3, 64, 128) and v["res0/bn0/mean_ema"].shape == (64,)
    assert {"res15/bn2/gamma", "res15/bn3/var_ema", "res15/W3", "conv1/bn/beta", "fc1/W", "outputTriplet/b"} <= set(v)
    assert v["fc1/W"].shape[1] == 500                       # TF stores FC matrices [in, out]
    W.save_tf_checkpoint(str(tmp_path / "ReID_general_weights"), v)
    back = W.load_any(str(tmp_path / "ReID_general_weights"), "reid")
    assert set(back) == set(w)
    for k, a in w.items():
        if isinstance(a, dict):
            assert all(torch.equal(a[q], back[k][q]) for q in a)
        else:
            assert torch.equal(a, back[k]), k


def test_oracle_forward_shapes_and_add_reid():
    units = [R.UNITS[0], ("res3", 2, (64, 64), (3, 3), (2, 1)), ("res15", 3, (32, 64, 96), (1, 3, 1), (1, 2, 1))]
    w = R.synth_weights(0, units)
    img = np.random.default_rng(1).integers(0, 256, (90, 140, 3), dtype=np.uint8)
    props = [{"bbox": [10.0, 12.0, 50.0, 40.0]}, {"bbox": [100.0, 30.0, 39.0, 58.0]}]
    out = R.add_reid(w, img, [dict(p) for p in props], units)
    assert len(out[0]["ReID"]) == 128 and isinstance(out[0]["ReID"][0], float)
    assert np.abs(np.array(out[0]["ReID"]) - np.array(out[1]["ReID"])).max() > 1e-3
    assert R.add_reid(w, img, [], units) == []
