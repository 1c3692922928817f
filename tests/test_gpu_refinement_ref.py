"""The HIP DeepLabv3+ / Xception-65 body against the fixture produced by EXECUTING the reference's graph code
(tools/make_golden_deeplab.py; tests/test_cpu_refinement_ref.py says what that pins): one [1,385,385,4] input, 2 middle units."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HR = json.load(open(os.path.join(GOLD, "deeplab_host_refs.json")))
REF = np.load(os.path.join(GOLD, "deeplab_ref.npz"))


def _close(a, b, tol):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() <= tol * max(1.0, float(np.abs(b).max()))


def test_whole_graph_hip_vs_reference_multi_scale_logits():
    from premvos_amd import synth
    from premvos_amd.refinement import RefinementNet
    nm = HR["num_middle"]
    x = np.random.default_rng(17).random((1, HR["size"], HR["size"], 4), dtype=np.float32)
    x[..., 3] = (x[..., 3] > 0.5)
    images = (x * 255).astype(np.float32)
    net = RefinementNet(synth.refinement_weights(7, nm), nm, use_graph=False)
    p = net.plan(1, 480, 854)
    # what the crop kernel leaves in the plan's input buffer: the net input after DeepLabV3Plus.py:12-14 and
    # feature_extractor.py:114-116, i.e. (2/255) * images - 1
    p.net_in.buf[...] = torch.from_numpy(np.float32(2.0 / 255.0) * images - 1.0).cuda()
    p.run([s for s in p.steps if s[0] not in ("refine_input", "refine_output")])
    torch.cuda.synchronize()
    nhwc = lambda t: t.torch().cpu().permute(0, 2, 3, 1).numpy()                     # noqa: E731
    assert _close(nhwc(p.xception_out)[:, :, :, ::16], REF["xception_out_sub"], 1e-3)
    assert _close(nhwc(p.aspp_out)[:, :, :, ::2], REF["aspp_sub"], 1e-3)
    assert _close(nhwc(p.decoder_out)[:, ::4, ::4, ::8], REF["decoder_sub"], 1e-3)
    assert _close(nhwc(p.logits), REF["logits"], 1e-3)


def test_refine_output_kernel_on_reference_segmentation_softmax():
    """premvos_refine_output_f32 against SegmentationSoftmax's eval branch executed by tools/make_golden_deeplab.py
    (SegmentationOutputLayers.py:17-135): frame-size masks / posteriors of five crop boxes + the forwarder's conf score."""
    from premvos_amd import _lib, ops
    ref = np.load(os.path.join(GOLD, "deeplab_ref_output.npz"))
    h, w = (int(v) for v in ref["frame_hw"])
    crops = ref["crops"]
    P = len(crops)
    li = ops.NHWC.alloc(P, 97, 97, 2)
    for i in range(P):
        li.buf[i, :, :, :2] = torch.from_numpy(ref[f"logits{i}"]).cuda()
    cr = torch.from_numpy(crops.astype(np.int32)).cuda()
    cnt = torch.tensor([P], dtype=torch.int32, device="cuda")
    mask = torch.zeros((P, h, w), dtype=torch.uint8, device="cuda")
    post = torch.zeros((P, h, w), device="cuda")
    conf = torch.zeros((P,), device="cuda")
    lib = _lib.load()
    ws = torch.zeros((int(lib.premvos_refine_output_workspace_bytes(P, 385, h, w)) + 3) // 4, device="cuda")
    _lib.check(lib.premvos_refine_output_f32(li.ptr, li.ps, 97, 97, cr.data_ptr(), cnt.data_ptr(), P, 385, h, w, mask.data_ptr(),
                                             post.data_ptr(), conf.data_ptr(), ws.data_ptr(), _lib.current_stream()))
    for i in range(P):
        want_m = np.unpackbits(ref[f"mask{i}"])[:h * w].reshape(h, w)
        want_p = ref[f"post{i}"]
        gp, gm = post[i].cpu().numpy(), mask[i].cpu().numpy()
        assert np.abs(gp - want_p).max() < 1e-5, i
        diff = gm != want_m
        assert not diff.any(), (i, int(diff.sum()))
        c = want_p.copy()                                   # FewShotSegmentationForwarder.py:144-148 on the reference's outputs
        c[want_m == 0] = 1 - want_p[want_m == 0]
        assert abs(float(conf[i]) - float((2 * c - 1).mean())) < 1e-5
