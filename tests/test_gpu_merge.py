"""GPU parity of the MergeTrack-side mask helpers (premvos_amd/mergetrack.py) with oracle/merge_oracle.py and
premvos_amd/rle.py -- integer work, so everything is bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import merge_oracle as M  # noqa: E402
from premvos_amd import rle  # noqa: E402


def _masks(seed, n, h, w, blobs=True):
    rng = np.random.default_rng(seed)
    out = np.zeros((n, h, w), np.uint8)
    yy, xx = np.mgrid[:h, :w]
    for i in range(n):
        if blobs:
            for _ in range(3):
                cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(2, max(h, w) / 3)
                out[i] |= ((yy - cy) ** 2 + (xx - cx) ** 2 < r * r).astype(np.uint8)
        else:
            out[i] = rng.random((h, w)) > 0.5
    return out


def _flow(seed, h, w, mag):
    rng = np.random.default_rng(seed)
    f = rng.normal(0, mag, (h, w, 2)).astype(np.float32)
    f[::3, ::5] = np.round(f[::3, ::5] * 64) / 64            # many exact 1/64 positions: the rounding ties of cvRound
    f[1::7, 2::3] = np.round(f[1::7, 2::3])                   # integer displacements
    return f


@pytest.mark.parametrize("h,w,mag,blobs", [(37, 53, 2.0, False), (64, 48, 40.0, True), (9, 300, 0.3, False)])
@pytest.mark.parametrize("binarize", [True, False])
def test_mask_warp_bit_exact(h, w, mag, blobs, binarize):
    from premvos_amd import mergetrack as MT
    m = _masks(h, 4, h, w, blobs)
    if not binarize:
        m = (m * np.random.default_rng(1).integers(1, 256, m.shape)).astype(np.uint8)      # general uint8 images
    f = _flow(w, h, w, mag)
    got = MT.warp_masks(m, f, binarize).cpu().numpy()
    for i in range(len(m)):
        assert np.array_equal(got[i], M.warp_flow(m[i], f, binarize)), i
    assert np.array_equal(MT.warp_flow(m[0], f, binarize), got[0])


def test_mask_warp_extreme_flows():
    from premvos_amd import mergetrack as MT
    m = _masks(5, 2, 20, 30, False)
    f = np.zeros((20, 30, 2), np.float32)
    f[:10] = 1e6
    f[10:, :15] = -1e6
    f[10:, 15:] = 0.999
    got = MT.warp_masks(m, f).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], M.warp_flow(m[i], f))
    assert got[:, :10].sum() == 0


def test_mask_iou_matches_oracle_and_pycocotools_conventions():
    from premvos_amd import mergetrack as MT
    a = _masks(1, 5, 45, 61)
    b = _masks(2, 3, 45, 61)
    a[4] = 0                                                   # empty proposal
    b[2] = 0                                                   # empty template
    got = MT.mask_iou(a, b)
    ref = M.mask_iou(list(a), list(b))
    assert got.shape == (5, 3) and got.dtype == np.float64 and np.array_equal(got, ref)
    assert got[4].sum() == 0 and got[:, 2].sum() == 0
    inter, aa, ab = MT.mask_overlap(a * 255, b)               # nonzero = foreground
    assert np.array_equal(aa.cpu().numpy(), a.reshape(5, -1).sum(1)) and np.array_equal(ab.cpu().numpy(), b.reshape(3, -1).sum(1))
    assert np.array_equal(MT.mask_iou(a, a).diagonal()[:4], np.ones(4))


@pytest.mark.parametrize("h,w", [(1, 1), (7, 5), (64, 33), (50, 4097)])
def test_rle_encode_matches_host_encoder(h, w):
    from premvos_amd import mergetrack as MT
    m = _masks(h + w, 5, h, w, blobs=h > 8)
    m[1] = 0
    m[2] = 1
    m[3] = np.random.default_rng(9).random((h, w)) > 0.5      # maximally ragged
    segs = MT.encode_masks(m)
    for i in range(5):
        assert segs[i] == rle.encode(m[i]), i
        assert np.array_equal(rle.decode(segs[i]), m[i])


def test_warp_proposals_matches_oracle(tmp_path):
    from premvos_amd import mergetrack as MT
    from premvos_amd.flow.driver import writeFlowFile
    m = _masks(11, 3, 48, 70)
    f = _flow(4, 48, 70, 3.0)
    props = [{"mask": m[i], "id": i + 1, "final_score": 0.1 * i, "object_score": 0.5 + 0.1 * i} for i in range(3)]
    ref = M.warp_proposals(props, f, rle)
    fn = str(tmp_path / "00000.flo")
    writeFlowFile(fn, f)
    for src in (fn, f, torch.from_numpy(f).cuda()):
        got = MT.warp_proposals(props, src)
        assert len(got) == 3
        for a, b in zip(got, ref):
            assert set(a) == set(b)
            assert np.array_equal(a["mask"], b["mask"]) and a["mask"].dtype == np.uint8
            assert a["segmentation"] == b["segmentation"] and a["bbox"] == b["bbox"]
            assert (a["score"], a["final_score"], a["object_score"], a["id"]) == (b["score"], b["final_score"], b["object_score"], b["id"])
    assert MT.warp_proposals([], f) == []


def test_full_size_frame_properties():
    """480x854, 20 masks: identity flow is the identity, IoU of a mask with itself is 1, RLE round-trips, and the
    device results agree with the oracle on a sample of masks."""
    from premvos_amd import mergetrack as MT
    m = _masks(3, 20, 480, 854)
    dm = torch.from_numpy(m).cuda()
    z = torch.zeros((480, 854, 2), device="cuda")
    assert torch.equal(MT.warp_masks(dm, z), dm)
    f = _flow(8, 480, 854, 6.0)
    w = MT.warp_masks(dm, f)
    for i in (0, 7, 19):
        assert np.array_equal(w[i].cpu().numpy(), M.warp_flow(m[i], f))
    iou = MT.mask_iou(w, dm)
    assert iou.shape == (20, 20) and np.array_equal(MT.mask_iou(dm, dm).diagonal(), np.ones(20))
    assert np.array_equal(iou[:3, :2], M.mask_iou([x for x in w[:3].cpu().numpy()], list(m[:2])))
    segs = MT.encode_masks(w)
    wn = w.cpu().numpy()
    for i in range(20):
        assert np.array_equal(rle.decode(segs[i]), wn[i])
    assert segs[5] == rle.encode(wn[5])


def _prop(mask, score=0.9, reid=None, **kw):
    d = {"segmentation": rle.encode(mask), "bbox": rle.to_bbox(rle.encode(mask)), "score": score,
         "ReID": np.zeros(128) if reid is None else reid}
    d.update(kw)
    return d


def test_merge_scores_selection_and_overlap_rules():
    """The host functions of the merge stage on hand-checkable inputs (merge_functions.py:38-76,96-149,243-248)."""
    from premvos_amd import mergetrack as MT
    a = np.zeros((8, 10), np.uint8); a[1:5, 1:5] = 1            # 16 px
    b = np.zeros((8, 10), np.uint8); b[3:7, 3:9] = 1            # 24 px, overlaps a in 2x2 = 4 px
    e1 = np.zeros(128); e1[0] = 5.0                             # |e1 - 0| = 5 -> ReID score 1 - 5/25 = 0.8
    templates = [_prop(a, score=1.0, id=1), _prop(b, score=0.75, reid=e1, id=2)]
    props = [_prop(a, score=0.9), _prop(b, score=0.6, reid=e1), _prop(np.zeros_like(a), score=0.5, reid=np.inf * np.ones(128))]
    s = MT.calculate_scores(props, templates)
    assert s.shape == (5, 2, 3)
    assert np.allclose(s[0], [[0.8, 0.2, 0.0]] * 2)                                        # objectness (score - .5) / .5
    assert np.allclose(s[1], [[1.0, 0.8, 0.0], [0.8, 1.0, 0.0]])                           # ReID; infinite distance -> 0
    assert np.allclose(s[2], [[0.2, 0.0, 1.0], [0.0, 0.2, 1.0]])                           # 1 - best OTHER template
    iou_ab = 4.0 / (16 + 24 - 4)
    assert np.allclose(s[3], [[1.0, iou_ab, 0.0], [0.5 * iou_ab, 0.5, 0.0]])               # warp IoU x (score-.5)/.5
    assert np.allclose(s[4], [[1 - 0.5 * iou_ab, 0.5, 1.0], [0.0, 1 - iou_ab, 1.0]])
    nw = MT.WEIGHTS / MT.WEIGHTS.sum()
    weighted = np.dot(nw, s.transpose((1, 0, 2)))
    obj = np.dot(np.array([1, 1]), s[:2].transpose((1, 0, 2)))
    sel = MT.calculate_selected_props(list(props), weighted, templates, MT.SCORE_THRESH, obj)
    assert [p["id"] for p in sel] == [1, 2] and sel[0]["segmentation"] == props[0]["segmentation"]
    assert sel[1]["segmentation"] == props[1]["segmentation"] and np.isclose(sel[0]["object_score"], 1.8)
    # nothing beats the threshold -> the empty proposal
    sel0 = MT.calculate_selected_props(list(props), np.full((2, 3), -1.0), templates, MT.SCORE_THRESH, obj)
    assert all(rle.area(p["segmentation"]) == 0 and p["final_score"] == MT.SCORE_THRESH for p in sel0)
    # overlap: the higher final score keeps the contested 2x2 block
    out = MT.remove_mask_overlap(sel)
    hi, lo = (0, 1) if sel[0]["final_score"] > sel[1]["final_score"] else (1, 0)
    assert out[hi]["mask"].sum() == (16, 24)[hi] and out[lo]["mask"].sum() == (16, 24)[lo] - 4
    assert (out[0]["mask"] & out[1]["mask"]).sum() == 0 and [p["id"] for p in out] == [1, 2]
    nxt = [dict(p, ReID=np.ones(128)) for p in out]
    upd = MT.update_templates(templates, nxt)
    assert np.array_equal(upd[1]["ReID"], e1) and upd[0]["id"] == 1 and np.array_equal(nxt[1]["ReID"], np.ones(128))


def test_palette_png_round_trip(tmp_path):
    from PIL import Image
    from premvos_amd import mergetrack as MT
    lab = np.zeros((6, 7), np.uint8); lab[1:3, 1:4] = 1; lab[3:5, 2:6] = 7
    MT.save_with_pascal_colormap(str(tmp_path / "a.png"), lab)
    im = Image.open(tmp_path / "a.png")
    assert im.mode == "P" and np.array_equal(np.array(im), lab)
    pal = np.array(im.getpalette()[:24]).reshape(8, 3)
    assert pal[1].tolist() == [128, 0, 0] and pal[2].tolist() == [0, 128, 0] and pal[7].tolist() == [128, 128, 128]
    t = MT.read_ann(str(tmp_path / "a.png"))
    assert [x["id"] for x in t] == [1, 7] and t[0]["bbox"].tolist() == [1.0, 1.0, 3.0, 2.0] and t[1]["score"] == 1.0
    MT.save_pngs([{"mask": (lab == 1).astype(np.uint8), "id": 3}, {"mask": (lab == 7).astype(np.uint8), "id": 5}],
                 str(tmp_path / "out" / "b.png"))
    assert set(np.unique(np.array(Image.open(tmp_path / "out" / "b.png")))) == {0, 3, 5}
