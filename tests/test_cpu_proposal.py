"""CPU tests of the proposal oracle + host logic (no GPU)."""
import json

import numpy as np
import pytest
import torch

from oracle import proposal_oracle as P

# utils/generate_anchors.py:20-38: the table in the reference's comments is the 1-based (MATLAB) form of the
# py-faster-rcnn anchors; the Python function returns it shifted by -1 (base window (0,0,15,15)).
REF_TABLE_1BASED = np.array([[-83, -39, 100, 56], [-175, -87, 192, 104], [-359, -183, 376, 200],
                             [-55, -55, 72, 72], [-119, -119, 136, 136], [-247, -247, 264, 264],
                             [-35, -79, 52, 96], [-79, -167, 96, 184], [-167, -343, 184, 360]], np.float64)


def test_generate_anchors_known_answer():
    assert np.array_equal(P.generate_anchors(), REF_TABLE_1BASED - 1)


def test_cell_anchor_field():
    ca = P.cell_anchors()
    assert ca.shape == (15, 4) and ca.dtype == np.float32
    # NUM_RATIO x NUM_SCALE layout (data.py:43), sqrt-areas 32..512, x2/y2 +1 => w = round-sized
    w, h = ca[:, 2] - ca[:, 0], ca[:, 3] - ca[:, 1]
    assert np.allclose(np.sqrt(w[5:10] * h[5:10]), [32, 64, 128, 256, 512])
    aa = P.all_anchors(3, 4)
    assert aa.shape == (3, 4, 15, 4)
    assert np.array_equal(aa[2, 3] - aa[0, 0], np.tile([48, 32, 48, 32], (15, 1)))
    from premvos_amd.proposal import cell_anchors
    assert np.array_equal(cell_anchors(), ca)


def test_custom_resize_shape_davis():
    from premvos_amd.proposal import custom_resize_shape
    for hw in ((480, 854), (1080, 1920), (480, 480), (854, 480), (100, 3000)):
        assert custom_resize_shape(*hw) == P.custom_resize_shape(*hw)
    assert P.custom_resize_shape(480, 854) == (749, 1333)
    assert P.custom_resize_shape(1080, 1920) == (750, 1333)


def test_decode_identity_and_clip():
    a = np.array([[10, 20, 50, 80]], np.float32)
    assert np.allclose(P.decode_bbox_target(np.zeros((1, 4), np.float32), a), a)
    big = P.decode_bbox_target(np.array([[0, 0, 100, 100]], np.float32), a)
    assert np.isfinite(big).all() and big[0, 2] - big[0, 0] == pytest.approx(40 * 1333 / 16, rel=1e-5)
    assert np.array_equal(P.clip_boxes(np.array([[-5, -5, 700, 900]], np.float32), 480, 640), [[0, 0, 640, 480]])


def test_nms_semantics():
    boxes = np.array([[0, 0, 10, 10], [0, 0, 10, 10.5], [20, 20, 30, 30], [0, 0, 10, 10]], np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.9], np.float32)
    assert P.nms_tf(boxes, scores, 10, 0.7) == [0, 2]          # ties -> lower index first; IoU>0.7 suppressed
    assert P.nms_tf(boxes, scores, 1, 0.7) == [0]
    assert P.iou_tf(boxes[0], np.array([5, 5, 5, 9], np.float32)) == 0     # zero-area box
    # IoU exactly at the threshold is NOT suppressed (strict >)
    a, b = np.array([0, 0, 2, 1], np.float32), np.array([0, 0, 1, 1], np.float32)
    assert P.iou_tf(a, b) == np.float32(0.5) and P.nms_tf(np.stack([a, b]), np.array([1, .5], np.float32), 5, 0.5) == [0, 1]


def test_topk_tie_rule():
    s = np.array([1, 3, 3, 2, 3, 0], np.float32)
    assert list(P.topk_indices(s, 2)) == [1, 2] and list(P.topk_indices(s, 4)) == [1, 2, 4, 3]


def test_roi_align_constant_and_linear_field():
    fm = torch.ones((1, 2, 10, 12))
    r = P.roi_align(fm, np.array([[1.0, 1.0, 9.0, 7.0]], np.float32), 14)
    assert torch.allclose(r, torch.ones_like(r))
    ramp = torch.arange(12, dtype=torch.float32).view(1, 1, 1, 12).expand(1, 1, 10, 12).contiguous()
    r = P.roi_align(ramp, np.array([[2.0, 2.0, 9.0, 8.0]], np.float32), 14)
    # bin centres of a linear field: x0 + (i+0.5)*w/14 - 0.5
    expect = 2.0 + (np.arange(14) + 0.5) * 7.0 / 14 - 0.5
    assert np.allclose(r[0, 0, 3].numpy(), expect, atol=1e-5)


def test_tail_threshold_and_empty():
    props = np.array([[0, 0, 50, 50], [100, 100, 180, 160]], np.float32)
    cls = np.array([[2.0, -2.0], [-1.0, 3.0]], np.float32)
    box = np.zeros((2, 1, 4), np.float32)
    fb, fp, fl, fi = P.fastrcnn_tail(cls, box, props, 300, 300)
    assert list(fi) == [1] and np.allclose(fb[0], props[1]) and list(fl) == [1]
    e = P.fastrcnn_tail(np.zeros((0, 2), np.float32), np.zeros((0, 1, 4), np.float32), np.zeros((0, 4), np.float32), 9, 9)
    assert e[0].shape == (0, 4)


def test_results_to_json_format():
    js = P.results_to_json(np.array([[10.26, 20.04, 110.31, 220.49]]), np.array([0.987]))
    assert js == [{"bbox": [10.3, 20.0, 100.0, 200.5], "score": 0.99}]
    json.dumps(js)
    from premvos_amd.proposal.driver import SecondDetectionResult, convert_results_to_json
    r = SecondDetectionResult(np.array([10.26, 20.04, 110.31, 220.49]), 0.987, 1, None, None, 1, None, None)
    assert convert_results_to_json([r]) == js


def test_small_oracle_forward_runs():
    w = P.synth_weights(0, (1, 1, 1, 1))
    img = np.random.default_rng(0).integers(0, 256, (64, 96, 3), dtype=np.uint8)
    (fb, fp, fl, fi), inter = P.model_forward(w, img, (1, 1, 1, 1), intermediates=True)
    assert inter["featuremap"].shape == (1, 1024, 4, 6)
    assert fb.shape[0] == fp.shape[0] <= 20 and inter["proposals"].shape[0] <= 100


# ---------------------------------------------------------------------------------------------------------------------
# property tests ("restate carefully, test tie cases with hypothesis")
from hypothesis import given, settings, strategies as st  # noqa: E402


def _iou_exact(a, b):
    """IoU of two boxes with integer corners in exact rational arithmetic (as a float64 quotient of integers)."""
    iw, ih = min(a[2], b[2]) - max(a[0], b[0]), min(a[3], b[3]) - max(a[1], b[1])
    if iw <= 0 or ih <= 0:
        return 0.0
    inter = iw * ih
    return inter / float((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


@settings(max_examples=60, deadline=None, derandomize=True)
@given(st.integers(0, 2 ** 31), st.integers(2, 40), st.sampled_from([0.3, 0.5, 0.7]))
def test_nms_and_topk_against_an_independent_greedy_formulation_with_ties(seed, n, thresh):
    """tf.image.non_max_suppression as the reference calls it (model.py:205-212): candidates in descending score order, ties
    to the lower index; a candidate is kept unless its IoU with an already kept box EXCEEDS the threshold.  Integer corners and
    scores from a four-value set make every comparison exact, so an independently written greedy loop must agree index for index."""
    rng = np.random.default_rng(seed)
    x0, y0 = rng.integers(0, 30, n), rng.integers(0, 30, n)
    boxes = np.stack([x0, y0, x0 + rng.integers(1, 25, n), y0 + rng.integers(1, 25, n)], 1).astype(np.float32)
    if n > 4:
        boxes[n // 2] = boxes[0]                                   # exact duplicates (IoU 1) and
        boxes[n - 1] = boxes[1]
    scores = rng.choice(np.array([0.9, 0.5, 0.5, 0.1], np.float32), n)      # many equal scores
    order = sorted(range(n), key=lambda i: (-float(scores[i]), i))
    assert list(P.topk_indices(scores, n)) == order
    keep = []
    for i in order:
        if all(_iou_exact(boxes[i], boxes[j]) <= thresh for j in keep):
            keep.append(i)
    got = P.nms_tf(boxes[:, [1, 0, 3, 2]], scores, n, thresh)
    # exact rational IoU vs the fp32 quotient of the TF kernel: identical decisions unless an IoU equals the threshold to 1e-6
    close = any(abs(_iou_exact(boxes[i], boxes[j]) - thresh) < 1e-6 for i in range(n) for j in range(i))
    assert close or list(got) == keep
    k = max(1, n // 3)
    assert list(P.nms_tf(boxes[:, [1, 0, 3, 2]], scores, k, thresh)) == keep[:k] or close
