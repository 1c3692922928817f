"""The FramePipeline object bench.py times (flow + proposal x2 + refinement, concurrent streams, grouped refinement):
results equal the stage objects run one by one, and are bit-identical from run to run (fixed-order reductions, frozen
tuning choices, no atomics on floats)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import proposal_oracle as PO  # noqa: E402
from oracle import pwc_oracle as O  # noqa: E402
from oracle import refinement_oracle as RO  # noqa: E402

BLOCKS, MIDDLE, B, P, H, W = (1, 1, 2, 1), 1, 3, 4, 96, 160


def _inputs():
    rng = np.random.default_rng(0)
    fa, fb = [], []
    for i in range(B):
        pair = O.synth_frame_pair(H, W, seed=7 + i, shift=(1.0 + 0.5 * i, -0.5))
        fr = (pair[0].permute(1, 2, 0) * 255).round().to(torch.uint8)
        fa.append(fr[..., :3])
        fb.append(fr[..., 3:])
    boxes = np.zeros((B, P, 4), np.float32)
    for b in range(B):
        for i in range(P):
            y0, x0 = rng.uniform(0, H - 40), rng.uniform(0, W - 60)
            boxes[b, i] = [y0, x0, y0 + rng.uniform(15, 40), x0 + rng.uniform(20, 60)]
    return torch.stack(fa).cuda(), torch.stack(fb).cuda(), torch.from_numpy(boxes).cuda()


@pytest.mark.parametrize("group", ["1", "2"])
def test_pipeline_equals_stages_and_is_deterministic(group, monkeypatch):
    monkeypatch.setenv("PREMVOS_REFINE_GROUP", group)
    from premvos_amd.pipeline import FramePipeline
    from premvos_amd.refinement import RefinementNet
    wf, wg, ws, wr = O.synth_state_dict(0), PO.synth_weights(0, BLOCKS), PO.synth_weights(1, BLOCKS), RO.synth_weights(0, MIDDLE)
    pipe = FramePipeline(wf, wg, ws, wr, batch=B, boxes_per_frame=P, num_blocks=BLOCKS, num_middle=MIDDLE)
    fa, fb, boxes = _inputs()
    keys = ("flow", "masks", "conf", "general_boxes", "general_probs", "general_count", "specific_boxes", "specific_probs",
            "specific_count")
    r1 = {k: v.clone() for k, v in pipe.step(fa, fb, boxes).items() if k in keys}
    torch.cuda.synchronize()
    for _ in range(2):
        r2 = pipe.step(fa, fb, boxes)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(r1[k], r2[k]), k                       # bit-identical replays
    assert int(r1["general_count"].min()) >= 0 and r1["flow"].shape == (B, H, W, 2) and r1["masks"].shape == (B, P, H, W)
    assert torch.isfinite(r1["flow"]).all() and int(r1["masks"].sum()) > 0
    # the same frames through a stand-alone refinement net, box by box
    net = RefinementNet(wr, MIDDLE)
    for b in range(B):
        p = net.refine(fa[b], boxes[b], max_boxes=P, with_posterior=True)
        diff = p.mask != r1["masks"][b]
        assert not bool(diff.any()) or (p.posterior[diff] - 0.5).abs().max().item() < 1e-4
        assert (p.conf - r1["conf"][b]).abs().max().item() < 1e-5
    # and the serial (one stream) pipeline gives the same numbers as the concurrent one
    serial = FramePipeline(wf, wg, ws, wr, batch=B, boxes_per_frame=P, num_blocks=BLOCKS, num_middle=MIDDLE, concurrent=False)
    r3 = serial.step(fa, fb, boxes)
    torch.cuda.synchronize()
    for k in keys:
        assert torch.equal(r1[k], r3[k]), k
