"""GPU parity of the ReID embedding path (premvos_amd/reid) with oracle/reid_oracle.py.
Tolerances: crops 1e-6 (same fp32 formula), feature maps / embeddings 1e-3 relative to the tensor's max."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import reid_oracle as R  # noqa: E402

SMALL_O = [R.UNITS[0], ("res3", 2, (64, 64), (3, 3), (2, 1)), ("res12", 2, (32, 64), (3, 3), (1, 2)),
           ("res15", 3, (32, 64, 96), (1, 3, 1), (1, 2, 1)), ("res16", 3, (48, 96, 128), (1, 3, 1), (1, 1, 1))]
SMALL = [(n, f, k, s) for n, _, f, k, s in SMALL_O]
BOXES = [[10.5, 20.0, 60.0, 40.5], [150.0, 80.0, 80.0, 60.0], [0.0, 0.0, 8.0, 30.0], [30.0, 5.0, 100.0, 110.0]]


def _img(seed, h=120, w=200):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.mark.parametrize("feed", [True, False])
def test_crops_match_oracle(feed):
    from premvos_amd import _lib
    from premvos_amd.reid import context_boxes
    img = _img(0)
    cb = context_boxes(BOXES, 120, 200, feed)
    out = torch.zeros((len(cb), 128, 128, 4), device="cuda")
    dimg, dcb = torch.from_numpy(img).cuda(), torch.from_numpy(cb).cuda()          # keep the device buffers alive
    _lib.check(_lib.load().premvos_reid_input_u8(dimg.data_ptr(), 120, 200, dcb.data_ptr(), len(cb), 128, int(feed),
                                                 out.data_ptr(), _lib.current_stream()), "reid_input")
    got = out.cpu().numpy()
    assert np.abs(got[..., 3]).max() == 0
    for i, b in enumerate(cb):
        assert np.abs(got[i, ..., :3] - R.make_crop(img, b, feed)).max() < 2e-6, i


def test_scale_shift_relu():
    from premvos_amd import _lib
    x = torch.randn((3, 5, 7, 24), generator=torch.Generator().manual_seed(0)).cuda()
    s, t = torch.rand(24).cuda() + 0.5, torch.randn(24, generator=torch.Generator().manual_seed(1)).cuda()
    out = torch.zeros((3, 5, 7, 28), device="cuda")
    _lib.check(_lib.load().premvos_scale_shift_relu_f32(x.data_ptr(), 24, 105, 24, s.data_ptr(), t.data_ptr(), out.data_ptr(),
                                                        28, 1, _lib.current_stream()), "ssr")
    assert (out[..., :24] - torch.relu(x * s + t)).abs().max().item() < 1e-6          # the kernel contracts to one FMA
    assert out[..., 24:].abs().max().item() == 0


@pytest.mark.parametrize("use_graph", [False, True])
def test_reduced_net_matches_oracle(use_graph):
    from premvos_amd.reid import ReIDNet, context_boxes
    w = R.synth_weights(0, SMALL_O)
    img = _img(1)
    net = ReIDNet(w, units=SMALL, use_graph=use_graph)
    emb = net.embed(torch.from_numpy(img).cuda(), BOXES, max_boxes=6).cpu().numpy()
    cb = context_boxes(BOXES, 120, 200, True)
    inter = {}
    ref = R.forward(w, np.stack([R.make_crop(img, b, True) for b in cb]), SMALL_O, inter)
    p = net.plan(6, 120, 200, True)
    for name in ("res0", "res12", "res16"):
        g = p.unit_out[name].torch().cpu()[:len(BOXES)]
        assert (g - inter[name]).abs().max().item() < 1e-3 * max(1.0, inter[name].abs().max().item()), name
    g = p.pooled.torch().cpu()[:len(BOXES)]
    assert (g - inter["conv1"]).abs().max().item() < 1e-3 * max(1.0, inter["conv1"].abs().max().item())
    assert emb.shape == (4, 128) and np.abs(emb - ref).max() < 1e-3 * max(1.0, np.abs(ref).max())
    # a single box gives the same embedding as inside a batch (up to the k-split the other batch size picks)
    one = net.embed(torch.from_numpy(img).cuda(), BOXES[3:4]).cpu().numpy()
    assert np.abs(one[0] - emb[3]).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_full_depth_net_matches_oracle():
    from premvos_amd.reid import ReIDNet, context_boxes
    w = R.synth_weights(0)
    img = _img(2, 480, 854)
    boxes = [[100.0, 50.0, 200.0, 300.0], [600.0, 300.0, 250.0, 180.0]]
    net = ReIDNet(w)
    emb = net.embed(torch.from_numpy(img).cuda(), boxes).cpu().numpy()
    cb = context_boxes(boxes, 480, 854, True)
    ref = R.forward(w, np.stack([R.make_crop(img, b, True) for b in cb]))
    assert np.abs(emb - ref).max() < 1e-3 * max(1.0, np.abs(ref).max())
    fl = sum(net.plan(2, 480, 854, True).flops.values()) / 2
    assert 30e9 < fl < 45e9                      # ~35 GFLOP per crop


def test_driver_json_contract(tmp_path):
    """add_ReID (in-merge) and the batch stage: same proposals + 'ReID' lists; proposals whose RLE is empty keep no key;
    weights come from a TF-format checkpoint named in the config."""
    from PIL import Image
    from premvos_amd import rle, weights as W
    from premvos_amd.reid import ReIDEngine, ReIDNet, add_ReID, forward_directory
    from premvos_amd.reid import driver as D
    w = R.synth_weights(3, SMALL_O)
    W.save_tf_checkpoint(str(tmp_path / "ReID_general_weights"), W.reid_weights_to_tf(w))
    cfg = tmp_path / "live"
    cfg.write_text(json.dumps({"load": str(tmp_path / "ReID_general_weights"), "model": "Re-ID"}))
    c = D.Config(str(cfg))
    assert c.str("model") == "Re-ID"
    eng = ReIDEngine(ReIDNet(D.load_weights(c.str("load")), units=SMALL))
    img = _img(4, 100, 160)
    (tmp_path / "img" / "seq").mkdir(parents=True)
    Image.fromarray(img).save(tmp_path / "img" / "seq" / "00000.png")
    props = [{"bbox": [10.0, 12.0, 50.0, 40.0], "score": 0.9}, {"bbox": [100.0, 30.0, 39.0, 58.0], "score": 0.8}]
    out = add_ReID([dict(p) for p in props], str(tmp_path / "img" / "seq" / "00000.png"), eng)
    ref = R.add_reid(w, img, [dict(p) for p in props], SMALL_O)
    for a, b in zip(out, ref):
        assert len(a["ReID"]) == 128 and np.abs(np.array(a["ReID"]) - np.array(b["ReID"])).max() < 1e-3 * max(1, np.abs(b["ReID"]).max())
    json.dumps(out)
    # batch stage: boxes come from the masks' RLE; an empty mask is skipped
    m = np.zeros((100, 160), np.uint8)
    m[20:70, 30:90] = 1
    bprops = [{"bbox": [0, 0, 1, 1], "segmentation": rle.encode(m)}, {"bbox": [0, 0, 1, 1], "segmentation": rle.encode(m * 0)}]
    (tmp_path / "bb" / "seq").mkdir(parents=True)
    (tmp_path / "bb" / "seq" / "00000.json").write_text(json.dumps(bprops))
    Image.fromarray(img).save(tmp_path / "img" / "seq" / "00000.jpg", quality=100)
    n = forward_directory(eng, str(tmp_path / "img") + "/", str(tmp_path / "bb") + "/", str(tmp_path / "out") + "/")
    res = json.load(open(tmp_path / "out" / "seq" / "00000.json"))
    assert n == 1 and len(res) == 2 and len(res[0]["ReID"]) == 128 and "ReID" not in res[1]
    jpg = np.asarray(Image.open(tmp_path / "img" / "seq" / "00000.jpg").convert("RGB"))
    cb = R.context_boxes([rle.to_bbox(bprops[0]["segmentation"])], 100, 160, feed=False)
    ref = R.forward(w, np.stack([R.make_crop(jpg, cb[0], feed=False)]), SMALL_O)
    assert np.abs(np.array(res[0]["ReID"]) - ref[0]).max() < 1e-3 * max(1.0, np.abs(ref).max())


def test_full_depth_net_and_crops_vs_reference_executed_fixture():
    """The HIP ReID path against tests/golden/reid_ref.npz -- the reference's own Network.build_tower / DAVIS_Forward_Feed code
    executed by tools/make_golden_reid.py (tests/test_cpu_reid_ref.py says what that pins)."""
    import os
    from premvos_amd import _lib
    from premvos_amd.reid import ReIDNet, context_boxes
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ref = np.load(os.path.join(gold, "reid_ref.npz"))
    x = np.random.default_rng(5).standard_normal((3, 128, 128, 3)).astype(np.float32)
    assert abs(float(x.astype(np.float64).sum()) - ref["input_checksum"][0]) < 1e-6
    net = ReIDNet(R.synth_weights(5), use_graph=False)
    p = net.plan(3, 96, 150, True)
    p.net_in.buf[..., :3] = torch.from_numpy(x).cuda()
    for _, fn in p.steps[1:]:                        # everything after the crop kernel: the net on the fixture's input
        fn()
    torch.cuda.synchronize()
    for name in ("res0", "res5", "res11", "res14", "res15", "res16"):
        g = p.unit_out[name].torch().cpu().permute(0, 2, 3, 1).numpy()[:, ::3, ::3, ::32]
        want = ref["act_" + name]
        assert g.shape == want.shape and np.abs(g - want).max() < 1e-3 * max(1.0, np.abs(want).max()), name
    emb = p.embeddings.cpu().numpy()
    assert np.abs(emb - ref["embedding"]).max() < 1e-3 * max(1.0, np.abs(ref["embedding"]).max())
    # crops: DAVISForwardFeedDataset._create_inputs_for_eval
    frame, boxes = ref["crop_frame"], ref["crop_boxes_xywh"]
    cb = context_boxes(boxes, 96, 150, True)
    assert np.array_equal(cb, ref["crop_context_boxes"].astype(np.int32))
    out = torch.zeros((len(cb), 128, 128, 4), device="cuda")
    dimg, dcb = torch.from_numpy(frame).cuda(), torch.from_numpy(cb).cuda()
    _lib.check(_lib.load().premvos_reid_input_u8(dimg.data_ptr(), 96, 150, dcb.data_ptr(), len(cb), 128, 1, out.data_ptr(),
                                                 _lib.current_stream()), "reid_input")
    got = out.cpu().numpy()[..., :3]
    assert np.abs(got[:, ::3, ::3] - ref["crops_sub"]).max() < 3e-6
    assert np.abs(got.mean(axis=(1, 2), dtype=np.float64) - ref["crops_mean"]).max() < 1e-5
    # batch stage: SimilarityDataset._load_crop_helper (excess >= 0, convert_image_dtype scaling)
    frame, boxes = ref["sim_frame"], ref["sim_boxes_xywh"]
    h, w = frame.shape[:2]
    cb = context_boxes(boxes, h, w, False)
    out = torch.zeros((len(cb), 128, 128, 4), device="cuda")
    dimg, dcb = torch.from_numpy(frame).cuda(), torch.from_numpy(cb).cuda()
    _lib.check(_lib.load().premvos_reid_input_u8(dimg.data_ptr(), h, w, dcb.data_ptr(), len(cb), 128, 0, out.data_ptr(),
                                                 _lib.current_stream()), "reid_input")
    got = out.cpu().numpy()[..., :3]
    assert np.abs(got[:, ::3, ::3] - ref["sim_crops_sub"]).max() < 3e-6
    assert np.abs(got.mean(axis=(1, 2), dtype=np.float64) - ref["sim_crops_mean"]).max() < 1e-5
