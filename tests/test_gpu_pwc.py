"""GPU parity of the full PWC-DC-Net forward against (a) golden vectors produced by importing the
reference PWCNet.py and (b) the CPU oracle on fresh seeded inputs.  Tolerance: fp32 flow, max-abs
1e-3 px relative to max(1,|flow|) (north_star: 'within a stated fp32 tolerance')."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pwc_oracle as O  # noqa: E402

TOL = 1e-3


def _net(seed, use_graph):
    from premvos_amd.flow import pwc_dc_net
    net = pwc_dc_net(None, use_graph=use_graph).cuda().eval()
    net.load_state_dict(O.synth_state_dict(seed))
    return net


@pytest.mark.parametrize("tag", ["64x64", "128x192"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_pwc_matches_reference_golden(golden_dir, tag, use_graph):
    g = np.load(os.path.join(golden_dir, f"pwc_{tag}.npz"))
    h, w = map(int, tag.split("x"))
    x = O.synth_frame_pair(h, w, seed=int(g["fseed"]), shift=tuple(float(s) for s in g["shift"]))
    net = _net(int(g["wseed"]), use_graph)
    got = net(x.cuda()).cpu().numpy()
    ref = g["flow2"]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < TOL * max(1.0, np.abs(ref).max())
    plan = net.plan(1, h, w)
    for lvl in (6, 5, 4, 3, 2):
        lf = plan.level_flow[lvl].torch().cpu().numpy()
        assert np.abs(lf - g[f"flow_l{lvl}"]).max() < TOL * max(1.0, np.abs(g[f"flow_l{lvl}"]).max()), lvl
    c16 = plan.feats[6].images(0, 1).torch().cpu().numpy()
    c26 = plan.feats[6].images(1, 1).torch().cpu().numpy()
    assert np.abs(c16 - g["c16"]).max() < TOL * max(1.0, np.abs(g["c16"]).max())
    assert np.abs(c26 - g["c26"]).max() < TOL * max(1.0, np.abs(g["c26"]).max())


def test_pwc_matches_oracle_batch2_and_replay():
    torch.manual_seed(0)
    sd = O.synth_state_dict(3)
    x = torch.cat([O.synth_frame_pair(128, 192, seed=5, shift=(2.0, 1.0)),
                   O.synth_frame_pair(128, 192, seed=6, shift=(-1.25, 0.5))], 0)
    with torch.no_grad():
        ref = O.pwc_forward(sd, x)
    net = _net(3, True)
    a = net(x.cuda()).cpu()
    b = net(x.cuda()).cpu()          # graph replay: identical
    assert torch.equal(a, b)
    assert (a - ref).abs().max().item() < TOL * max(1.0, ref.abs().max().item())
    # zero motion / identical frames also go through
    z = x.clone()
    z[:, 3:] = z[:, :3]
    with torch.no_grad():
        refz = O.pwc_forward(sd, z)
    assert (net(z.cuda()).cpu() - refz).abs().max().item() < TOL * max(1.0, refz.abs().max().item())


def test_pwc_rejects_bad_shapes():
    net = _net(0, False)
    with pytest.raises(ValueError):
        net(torch.zeros((1, 6, 60, 64), device="cuda"))


@pytest.mark.parametrize("h,w", [(100, 150), (128, 192), (97, 250)])
def test_flow_stage_pre_post_and_end_to_end(h, w, tmp_path):
    """uint8 frames -> .flo payload: bit-exact fixed-point resize, 1e-6 float resize, end-to-end flow
    within TOL of the oracle chain (script_pwc_multi.py:33-70 restated in oracle/cv_resize_oracle.py)."""
    from oracle import cv_resize_oracle as R
    from premvos_amd import _lib
    from premvos_amd.flow.driver import FlowStage, readFlowFile, writeFlowFile
    rng = np.random.default_rng(h * w)
    base = O.synth_frame_pair(h + (-h) % 8, w + (-w) % 8, seed=h)[0].numpy()
    im1 = (base[:3, :h, :w].transpose(1, 2, 0) * 255).round().astype(np.uint8)
    im2 = (base[3:, :h, :w].transpose(1, 2, 0) * 255).round().astype(np.uint8)
    im2[::7, ::5] = rng.integers(0, 256, im2[::7, ::5].shape, dtype=np.uint8)     # hard edges for the resize
    sd = O.synth_state_dict(2)
    stage = FlowStage(sd, batch=2)
    a = torch.from_numpy(np.stack([im1, im2])).cuda()
    b = torch.from_numpy(np.stack([im2, im1])).cuda()
    got = stage.run(a, b).cpu().numpy()
    x0, h_, w_ = R.flow_preprocess(im1, im2)
    # (1) preprocess kernel: bit-exact vs the restated cv2 fixed-point resize
    img = stage.plan.img.buf.cpu().numpy()            # [4,h_,w_,4]: (im1,im2 | im2,im1)
    ref_a = x0[0, :3].transpose(1, 2, 0)
    ref_b = x0[0, 3:].transpose(1, 2, 0)
    assert np.array_equal(img[0, :, :, :3], ref_a) and np.array_equal(img[2, :, :, :3], ref_b)
    assert np.array_equal(img[1, :, :, :3], ref_b) and np.array_equal(img[3, :, :, :3], ref_a)
    assert not img[..., 3].any()
    # (2) postprocess kernel vs restated float resize on the GPU's own flow2
    f2 = stage.plan.flow2_nhwc.torch().cpu().numpy()
    for i in range(2):
        ref_post = R.flow_postprocess(f2[i], h, w, h_, w_)
        assert np.abs(got[i] - ref_post).max() < 1e-5 * max(1.0, np.abs(ref_post).max())
    # (3) end to end vs the oracle chain
    with torch.no_grad():
        ref_f2 = O.pwc_forward(sd, torch.from_numpy(x0))[0].numpy()
    ref = R.flow_postprocess(ref_f2, h, w, h_, w_)
    assert np.abs(got[0] - ref).max() < TOL * max(1.0, np.abs(ref).max()) * 20
    # (4) .flo round trip
    fn = str(tmp_path / "t.flo")
    writeFlowFile(fn, got[0])
    assert np.array_equal(readFlowFile(fn), got[0])
    assert _lib.lib_path().endswith("libpremvos_hip.so")


def test_kept_winograd_slabs_leave_the_flow_bit_identical(monkeypatch):
    """The bench's flow shape (16 pairs at 512x896, the shipped table's choices): the F(4x4) layers of the estimator levels keep one
    input-transform slab per level and transform only the channels the previous layer added (PWCNet.py:201-205) -- the flow and
    every level's concat buffer are bit-identical to the plan in which each layer transforms its whole window again."""
    x = torch.cat([O.synth_frame_pair(512, 896, seed=20 + i, shift=(1.5 * i - 3.0, 0.75 * i)) for i in range(4)], 0).cuda()
    x = x.repeat(4, 1, 1, 1)
    monkeypatch.setenv("PREMVOS_VSLAB", "0")
    net0 = _net(11, False)
    f0 = net0(x)
    p0 = net0.plan(16, 512, 896)
    assert p0.vslab is None and not p0.vslab_layers
    monkeypatch.setenv("PREMVOS_VSLAB", "1")
    net1 = _net(11, False)
    f1 = net1(x)
    p1 = net1.plan(16, 512, 896)
    torch.cuda.synchronize()
    kept = p1.vslab_layers
    assert len(kept) >= 8, kept                                          # levels 5 ... 2: three or four F(4x4) layers each
    assert kept["conv:conv3_1"] == (320, 128) and kept["conv:conv3_3"] == (96, 96) and kept["conv:conv2_2"] == (192, 128)
    assert torch.equal(f0, f1)
    for lvl in (5, 4, 3, 2):
        assert torch.equal(p0.xbufs[lvl].buf, p1.xbufs[lvl].buf), lvl
