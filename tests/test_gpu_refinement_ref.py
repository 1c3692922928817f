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
