"""oracle/merge_oracle.py (restated cv2.remap / pycocotools iou semantics) against analytic cases -- neither library
exists in this image, so these hand-derivable cases are the only pins (the oracle header says 'parity unpinned')."""
import numpy as np

from oracle import merge_oracle as M
from premvos_amd import rle


def _mask(seed, h=11, w=17):
    return (np.random.default_rng(seed).random((h, w)) > 0.5).astype(np.uint8)


def test_zero_and_integer_flows_are_exact_shifts():
    m = _mask(0)
    z = np.zeros(m.shape + (2,), np.float32)
    assert np.array_equal(M.warp_flow(m, z), m)
    f = z.copy()
    f[..., 0], f[..., 1] = 3, -2                       # out(y, x) = m(y + 2, x - 3), zero outside
    exp = np.zeros_like(m)
    exp[:-2, 3:] = m[2:, :-3]
    assert np.array_equal(M.warp_flow(m, f), exp)
    f[..., 0] = 1000.0                                  # everything comes from outside the image
    assert M.warp_flow(m, f).sum() == 0


def test_fractions_are_quantised_to_one_32nd_and_rounded_half_up():
    m = np.zeros((1, 4), np.uint8)
    m[0, 1] = 1
    f = np.zeros((1, 4, 2), np.float32)
    f[..., 0] = 0.5                                     # out(x) = (m(x-1) + m(x)) / 2 -> 0.5 rounds up to 1
    assert M.warp_flow(m, f, binarize=False).tolist() == [[0, 1, 1, 0]]
    f[..., 0] = 0.5 + 1.0 / 32                          # weights 15/32 | 17/32
    assert M.warp_flow(m, f, binarize=False).tolist() == [[0, 0, 1, 0]]
    f[..., 0] = 0.5 + 1.0 / 128                         # 0.5078 * 32 = 16.25 -> 16/32: same as 0.5
    assert M.warp_flow(m, f, binarize=False).tolist() == [[0, 1, 1, 0]]
    g = np.full((1, 4), 200, np.uint8)                  # general uint8 values: (sum w v + 2^14) >> 15
    f[..., 0] = 0.25
    assert M.warp_flow(g, f, binarize=False).tolist() == [[150, 200, 200, 200]]
    assert M.warp_flow(g, f, binarize=True).sum() == 0  # "== 1" of the reference's binarisation


def test_mask_iou_known_answers():
    a = np.array([[1, 1], [0, 0]], np.uint8)
    b = np.array([[1, 0], [1, 0]], np.uint8)
    e = np.zeros((2, 2), np.uint8)
    iou = M.mask_iou([a, b, e], [a, b, e])
    assert np.allclose(iou, [[1, 1 / 3, 0], [1 / 3, 1, 0], [0, 0, 0]])          # empty vs empty: i == 0 -> 0
    assert iou.dtype == np.float64 and iou.shape == (3, 3)


def test_warp_proposals_bookkeeping():
    m = _mask(3)
    f = np.zeros(m.shape + (2,), np.float32)
    f[..., 1] = 1
    props = [{"mask": m, "id": 7, "final_score": 0.4, "object_score": 0.9}]
    out = M.warp_proposals(props, f, rle)
    assert out[0]["id"] == 7 and out[0]["score"] == 0.5 * 1.4 and out[0]["final_score"] == 0.4
    assert np.array_equal(rle.decode(out[0]["segmentation"]), out[0]["mask"])
    assert out[0]["bbox"] == rle.to_bbox(rle.encode(out[0]["mask"]))


def test_oracle_vs_reference_executed_warp_proposals():
    """tests/golden/merge_ref.npz: MergeTrack/merge_functions.py's get_flow / warp_flow / warp_proposals EXECUTED by
    tools/make_golden_merge.py with a recording cv2.remap (integer-valued flows: bilinear == gather) and dense-mask stand-ins for
    pycocotools.  Pins the sampling map (-flow + grid), `== 1` binarisation and warp_proposals' bookkeeping."""
    import json
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ref = np.load(os.path.join(gold, "merge_ref.npz"))
    hr = json.load(open(os.path.join(gold, "merge_host_refs.json")))
    assert hr["interpolation_flag_is_INTER_LINEAR"] and hr["remap_calls"] == len(ref["masks"]) and hr["segmentation_counts_is_str"]
    flow = ref["flow"]
    assert np.array_equal(M.remap_map(flow), ref["remap_map"])
    for m, want in zip(ref["masks"], ref["warped_masks"]):
        assert np.array_equal(M.warp_flow(m, flow), want)
    assert np.array_equal(M.warp_flow(ref["grey"], flow, binarize=False), ref["grey_warped"])
    assert np.array_equal(M.warp_flow(ref["grey"], flow), ref["grey_warped_bin"])
    props = [{"mask": m, "id": w["id"], "final_score": w["final_score"], "object_score": w["object_score"]}
             for m, w in zip(ref["masks"], hr["warped"])]
    out = M.warp_proposals(props, flow, rle)
    assert sorted(out[0].keys()) == hr["warped_keys"]
    for o, w, bb in zip(out, hr["warped"], ref["warped_bbox"]):
        assert o["score"] == w["score"] and o["id"] == w["id"] and o["final_score"] == w["final_score"]
        assert np.array_equal(np.asarray(o["bbox"], np.float64), bb)
