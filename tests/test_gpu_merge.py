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


@pytest.mark.parametrize("n", [1, 7, 8, 210, 4097, 20 * 480 * 854])
def test_mask_bit_pack_round_trip(n):
    """premvos_mask_pack_bits_u8 / unpack: np.packbits(bitorder='little') semantics, any length, any non-zero = 1."""
    from premvos_amd.parallel import _hip_pack_bits, _hip_unpack_bits
    g = torch.Generator().manual_seed(n)
    m = (torch.rand(n, generator=g) > 0.5).to(torch.uint8) * torch.randint(1, 255, (n,), generator=g, dtype=torch.uint8)
    bits = torch.zeros((n + 7) // 8, dtype=torch.uint8, device="cuda")
    _hip_pack_bits(m.cuda(), bits)
    assert np.array_equal(bits.cpu().numpy(), np.packbits(m.numpy() != 0, bitorder="little"))
    back = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
    _hip_unpack_bits(bits, back)
    assert torch.equal(back.cpu(), (m != 0).to(torch.uint8))


def test_warp_proposals_with_resident_masks():
    """SURVEY 8(f1) residency: CUDA masks in, CUDA masks out, same RLE / bbox / scores as the host-mask path."""
    from premvos_amd import mergetrack as MT
    m = _masks(11, 3, 40, 56)
    f = _flow(12, 40, 56, 4.0)
    props = [{"mask": m[i], "id": i + 1, "final_score": 0.2 * i, "object_score": 0.5} for i in range(3)]
    host = MT.warp_proposals(props, f)
    dm = torch.from_numpy(m).cuda()
    dev = MT.warp_proposals([dict(p, mask=dm[i]) for i, p in enumerate(props)], torch.from_numpy(f).cuda(), device_masks=True)
    for a, b in zip(dev, host):
        assert isinstance(a["mask"], torch.Tensor) and a["mask"].is_cuda
        assert np.array_equal(a["mask"].cpu().numpy(), b["mask"])
        assert a["segmentation"] == b["segmentation"] and a["bbox"] == b["bbox"] and a["score"] == b["score"]
    again = MT.warp_proposals(dev, torch.from_numpy(f).cuda(), device_masks=True)          # the result feeds the next frame
    assert len(again) == 3 and again[0]["mask"].is_cuda


def test_mask_warp_vs_reference_executed_warp_proposals():
    """The HIP mask warp + warp_proposals against tests/golden/merge_ref.npz (merge_functions.py executed by
    tools/make_golden_merge.py; see tests/test_cpu_merge.py for what that pins)."""
    import json
    import os
    from premvos_amd import mergetrack as MT
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ref = np.load(os.path.join(gold, "merge_ref.npz"))
    hr = json.load(open(os.path.join(gold, "merge_host_refs.json")))
    flow = ref["flow"]
    assert np.array_equal(MT.warp_masks(ref["masks"], flow).cpu().numpy(), ref["warped_masks"])
    assert np.array_equal(MT.warp_flow(ref["grey"], flow, binarize=False), ref["grey_warped"])
    assert np.array_equal(MT.warp_flow(ref["grey"], flow), ref["grey_warped_bin"])
    props = [{"mask": m, "id": w["id"], "final_score": w["final_score"], "object_score": w["object_score"]}
             for m, w in zip(ref["masks"], hr["warped"])]
    out = MT.warp_proposals(props, flow)
    assert sorted(out[0].keys()) == hr["warped_keys"]
    for o, w, bb, wm in zip(out, hr["warped"], ref["warped_bbox"], ref["warped_masks"]):
        assert o["score"] == w["score"] and o["id"] == w["id"] and o["object_score"] == w["object_score"]
        assert np.array_equal(np.asarray(o["bbox"], np.float64), bb) and np.array_equal(np.asarray(o["mask"]), wm)
        assert isinstance(o["segmentation"]["counts"], str)


@pytest.mark.parametrize("shape", [(6, 9, 11, 9, 11), (5, 21, 30, 17, 26), (40, 480, 854, 480, 854), (3, 64, 40, 1, 1)])
def test_pooled_run_boundaries_and_the_host_string_packer(shape):
    """premvos_rle_boundaries_pooled_u8 (round 6: a chunk's masks -> ONE variable-length pool, on the rank that produced them; the
    h x w window of masks that sit in a larger block) + premvos_rle_strings_host: the strings are rle.encode's (the numpy reader of
    the same format), the per-mask kernel agrees, an all-zero and an all-one mask are covered, and a pool that is too small says so."""
    from premvos_amd import _lib, mergetrack as MT
    n, H, W, h, w = shape
    rng = np.random.default_rng(n * 1000 + h)
    big = np.zeros((n, H, W), np.uint8)
    blobs = _masks(n + 1, n, H, W) if min(H, W) >= 8 else (rng.random((n, H, W)) > 0.5).astype(np.uint8)
    big[:] = blobs * rng.integers(1, 255, (n, 1, 1)).astype(np.uint8)         # any non-zero byte is foreground
    big[0] = 0
    if n > 1:
        big[1] = 7
    if n > 2:
        big[2] = (rng.random((H, W)) > 0.5)                                    # a very ragged one
    m = torch.from_numpy(big).cuda()
    lib = _lib.load()
    cap = int(sum(len(rle.counts_from_mask(big[i, :h, :w])) for i in range(n))) + 8
    pool = torch.full((cap,), -1, dtype=torch.int32, device="cuda")
    off = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
    ws = torch.empty((int(lib.premvos_rle_workspace_bytes(n, h, w)) + 3) // 4, dtype=torch.int32, device="cuda")
    _lib.check(lib.premvos_rle_boundaries_pooled_u8(m.data_ptr(), n, h, w, H * W, W, pool.data_ptr(), cap, off.data_ptr(), ws.data_ptr(),
                                                    _lib.current_stream()), "pooled")
    o, p = off.cpu().numpy(), pool.cpu().numpy()
    assert o[0] == 0 and np.all(np.diff(o) >= 0) and o[-1] <= cap and np.all(p[o[-1]:] == -1)
    strings = rle.strings_from_pool(p, o, h * w)
    for i in range(n):
        win = big[i, :h, :w]
        assert strings[i] == rle.encode(win)["counts"], i
        pos = p[o[i]:o[i + 1]]
        assert np.all(np.diff(pos) > 0) if len(pos) > 1 else True
    if (H, W) == (h, w):                                                       # the per-mask kernel of round 2: same boundaries
        segs = MT.encode_masks(m)
        assert [s["counts"] for s in segs] == strings
    # a pool that cannot hold the chunk: offsets[-1] still reports the need, nothing is written past the capacity
    small = max(1, int(o[-1]) // 2)
    pool2 = torch.full((small + 4,), -1, dtype=torch.int32, device="cuda")
    _lib.check(lib.premvos_rle_boundaries_pooled_u8(m.data_ptr(), n, h, w, H * W, W, pool2.data_ptr(), small, off.data_ptr(), ws.data_ptr(),
                                                    _lib.current_stream()), "pooled")
    assert int(off[-1]) == int(o[-1]) and np.all(pool2.cpu().numpy()[small:] == -1)
    assert np.array_equal(pool2.cpu().numpy()[:min(small, int(o[-1]))], p[:min(small, int(o[-1]))])


def test_merge_ingest_tool_reproduces_the_one_rank_tree_at_small_size(tmp_path, monkeypatch):
    """tools/time_merge_ingest.py (bench.py's `merge_ingest` leg) end to end on small frames and reduced nets: three fake ranks,
    recorded buffers replayed alone and beside the real streaming driver, and the round-5 form of the merge side -- every tree
    byte-identical to the one-rank tree (``measure`` raises otherwise)."""
    import importlib.util
    import os
    from oracle import proposal_oracle as PO, pwc_oracle as O, refinement_oracle as RO
    from premvos_amd import stream
    spec = importlib.util.spec_from_file_location("time_merge_ingest", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools", "time_merge_ingest.py"))
    tmi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tmi)
    monkeypatch.setenv("PREMVOS_DRIVER_BATCH", "2")
    monkeypatch.setenv("PREMVOS_IO_WRITERS", "3")
    job = tmi.build_job(str(tmp_path / "job"), 7, 3, h=120, w=200, weights=False)      # 7 frames, chunks of 3: a ragged last chunk
    wd = tmp_path / "w"
    wd.mkdir()
    torch.save({"state_dict": O.synth_state_dict(0)}, wd / "pwc.pth.tar")
    torch.save(PO.synth_weights(0, (1, 1, 2, 1)), wd / "general.pt")
    torch.save(PO.synth_weights(1, (1, 1, 2, 1)), wd / "specific.pt")
    torch.save(RO.synth_weights(0, 1), wd / "refine.pt")
    sp = stream.StreamPipeline(str(wd / "pwc.pth.tar"), str(wd / "general.pt"), str(wd / "specific.pt"), str(wd / "refine.pt"),
                               batch=3, out=str(tmp_path / "out"))
    for legacy in (False, True):
        rep = tmi.measure(sp, job["clips"], 7, 3, legacy=legacy, tmp_out=str(tmp_path / f"t{int(legacy)}"))
        for mode in ("alone", "beside"):
            assert rep[mode]["byte_identical_trees"] and rep[mode]["files"] == 3 * (5 * 7 - 1), rep
            assert rep[mode]["rle_overflow_chunks"] == 0
        assert rep["writer_threads"] == (1 if legacy else 3)
