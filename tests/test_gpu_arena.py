"""Liveness-planned activation memory (premvos_amd/arena.py) on the GPU: a plan built on the packed arena gives the SAME BITS as
one that owns a buffer per tensor (PREMVOS_ARENA=0), in both arithmetic modes, eager and through a replayed HIP graph twice
(stale bytes of an earlier run in a re-used range must not reach any result); and the arena is much smaller."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import proposal_oracle as P  # noqa: E402
from oracle import refinement_oracle as R  # noqa: E402

SMALL = (2, 2, 3, 2)
BOXES = [[10.0, 20.0, 90.0, 150.0], [0.0, 0.0, 60.0, 70.0], [40.5, 33.2, 119.0, 199.0]]


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("mode_mask", [False, True])
def test_proposal_plan_on_the_arena_is_bit_identical(monkeypatch, precision, mode_mask):
    from premvos_amd.proposal import ProposalNet
    w = P.synth_weights(1, SMALL)
    imgs = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (2, 160, 256, 3), dtype=np.uint8)).cuda()
    res = {}
    for on in ("0", "1"):
        monkeypatch.setenv("PREMVOS_ARENA", on)
        net = ProposalNet(w, SMALL, use_graph=True, precision=precision, mode_mask=mode_mask)
        for rep in range(2):                 # the second replay runs on what the first left in every re-used range
            p = net.run_resized(imgs)
        torch.cuda.synchronize()
        res[on] = [t.clone() for t in (p.featuremap.buf, p.rpn_out.buf, p.rois, p.roi_idx, p.roi_count, p.feat5.buf, p.head.buf,
                                       p.final_boxes, p.final_probs, p.final_idx, p.final_count)]
        if mode_mask:
            res[on].append(p.final_masks.buf.clone())
        rep_ = p.arena.report()
        if on == "1":
            assert rep_["arena_bytes"] < 0.8 * rep_["one_buffer_per_tensor_bytes"], rep_          # (full depth at 749x1333: 0.22)
            assert rep_["arena_bytes"] <= 1.25 * rep_["peak_live_bytes"], rep_
        else:
            assert rep_["arena_bytes"] == 0
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_refinement_plan_on_the_arena_is_bit_identical(monkeypatch, precision):
    from premvos_amd.refinement import RefinementNet
    nm = 2
    w = R.synth_weights(1, nm)
    rng = np.random.default_rng(1)
    frames = torch.from_numpy(rng.integers(0, 256, (2, 120, 200, 3), dtype=np.uint8)).cuda()
    boxes = torch.tensor([BOXES, BOXES[::-1]]).cuda()
    res = {}
    for on in ("0", "1"):
        monkeypatch.setenv("PREMVOS_ARENA", on)
        net = RefinementNet(w, nm, use_graph=True, precision=precision)
        for rep in range(2):
            p = net.refine_group(frames, boxes, with_posterior=True)
        torch.cuda.synchronize()
        res[on] = [t.clone() for t in (p.xception_out.buf, p.aspp_out.buf, p.decoder_out.buf, p.logits.buf, p.mask_g, p.posterior_g, p.conf_g)]
        # the packed (eager) form shares the builder
        q = net.refine_packed(frames, [boxes[0, :2], boxes[1]], slots=6, max_frames=2)
        torch.cuda.synchronize()
        res[on] += [q.mask_g.clone(), q.conf_g.clone()]
        if on == "1":
            rep_ = p.arena.report()
            assert rep_["arena_bytes"] < 0.8 * rep_["one_buffer_per_tensor_bytes"], rep_
            assert rep_["arena_bytes"] <= 1.25 * rep_["peak_live_bytes"], rep_
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_plans_of_one_lane_interleaved_on_shared_bytes_keep_their_bits(monkeypatch, precision):
    """the refinement plans of one lane live in ONE shared buffer that is never re-zeroed.  Plan A, plan B (another slot
    count: other offsets for every tensor), plan A again -- through replayed graphs and the packed eager form -- with the arena
    poisoned (NaN bit patterns) before every run: the masks / conf / posterior bits are the ones of PREMVOS_ARENA=0, i.e. no
    kernel reads a byte that the SAME run has not written (crop slots at or beyond ``count`` and unused packed slots included)."""
    from premvos_amd.refinement import RefinementNet
    nm = 2
    w = R.synth_weights(1, nm)
    rng = np.random.default_rng(2)
    frames = torch.from_numpy(rng.integers(0, 256, (2, 120, 200, 3), dtype=np.uint8)).cuda()
    b3 = torch.tensor([BOXES, BOXES[::-1]]).cuda()                     # plan A: 2 frames x 3 boxes
    b5 = torch.tensor([BOXES + BOXES[:2], (BOXES + BOXES[:2])[::-1]]).cuda()   # plan B: 2 frames x 5 boxes
    cnt = torch.tensor([2, 3], dtype=torch.int32).cuda()              # frame 0 leaves a slot beyond its count

    def sequence(net, poison):
        out = []

        def fill():
            if poison:
                torch.cuda.synchronize()
                for kinds in net._lane_bytes.values():
                    for buf in kinds.values():
                        buf.view(torch.int32).fill_(0x7FC00001)      # a quiet NaN as fp32; {NaN, tiny} as a bf16 pair
        for step in ("A", "B", "A", "packed", "B", "packed-small", "A"):
            fill()
            if step == "A":
                p = net.refine_group(frames, b3, counts=cnt, with_posterior=True)
                for t in (p.mask_g, p.conf_g, p.posterior_g):          # (the valid slots: what a caller may read)
                    out += [t[0, :2].clone(), t[1, :3].clone()]
            elif step == "B":
                p = net.refine_group(frames, b5, with_posterior=True)
                out += [p.mask_g.clone(), p.conf_g.clone(), p.posterior_g.clone()]
            elif step == "packed":
                q = net.refine_packed(frames, [b3[0, :2], b5[1]], slots=8, max_frames=2)
                out += [q.mask_g[0, :7].clone(), q.conf_g[0, :7].clone()]
            else:
                q = net.refine_packed(frames[:1], [b3[0, :1]], slots=2, max_frames=2)
                out += [q.mask_g[0, :1].clone(), q.conf_g[0, :1].clone()]
            torch.cuda.synchronize()
        return out

    monkeypatch.setenv("PREMVOS_ARENA", "0")
    ref = sequence(RefinementNet(w, nm, use_graph=True, precision=precision), poison=False)
    monkeypatch.setenv("PREMVOS_ARENA", "1")
    net = RefinementNet(w, nm, use_graph=True, precision=precision)
    got = sequence(net, poison=True)
    assert len(net._lane_bytes[0]) >= 1 and len({id(p.arena.bufs["f32"]) for p in net._plans.values()}) <= 2   # the plans do share
    for i, (a, b) in enumerate(zip(ref, got)):
        if a.dtype.is_floating_point:
            assert not torch.isnan(b).any(), i
        assert torch.equal(a, b), i
