"""The oracle checks the object bench.py TIMES (VERDICT r03 next #1a): FramePipeline(batch = 16, full depth, the shipped
configuration table, concurrent streams, refinement groups of 8 frames on two lanes), built exactly as bench.py:348 builds it
-- premvos_amd.synth weights, the synthetic clip, the seeded boxes.  At batch 16 the table picks other kernels than at the
batch 1 of tests/test_gpu_fullsize.py (F(4x4,3x3) on every group2 / group3 conv2 + RPN, the streaming pointwise kernel, other
k-splits), so this is the test that pins THOSE choices: two of the 16 frames (one per refinement lane) go through oracle/*.

Bars (north_star): flow <= 1e-3 px (relative to max(1, |flow|)), feature maps / RPN logits <= 1e-3 / 2e-3 relative, the 100
RPN indices and the final detections' indices STRICTLY those model.py:169-217 / :438-491 select from the GPU's own logits and
head outputs, all 100 indices shared with the CPU net, refinement mask logits <= 1e-3 relative, conf_score <= 1e-3, mask pixels
may differ only where the oracle's posterior is within 2e-3 of 0.5.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cv_resize_oracle as CR  # noqa: E402
from oracle import proposal_oracle as PO  # noqa: E402
from oracle import pwc_oracle as O  # noqa: E402
from oracle import refinement_oracle as RO  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, P, H, W = 16, 20, 480, 854
FRAMES = (3, 12)            # one in each refinement group / lane (frames 0-7: lane 0, 8-15: lane 1)
BOXES = (0, 7, 19)          # refinement boxes checked per frame


def _families(descs):
    fam = {}
    for d in descs:
        k = {1: "direct", 2: "wino2x2_slab", 3: "wino2x2_fused", 4: "wino4x4", 5: "stream", 6: "bf16x3_s8"}.get(d.tile_hint, "igemm")
        fam[k] = fam.get(k, 0) + 1
    return fam


def test_bench_pipeline_object_against_the_oracle(monkeypatch):
    for k in ("PREMVOS_FORCE_KERNEL", "PREMVOS_PRECISION", "PREMVOS_REFINE_GROUP", "PREMVOS_REFINE_LANES", "PREMVOS_PIPELINE_SERIAL",
              "PREMVOS_AUTOTUNE", "PREMVOS_TUNE_TABLE"):
        monkeypatch.delenv(k, raising=False)
    from premvos_amd import ops, synth
    from premvos_amd.pipeline import FramePipeline
    wf, wg, ws, wr = synth.pwc_state_dict(0), synth.proposal_weights(0), synth.proposal_weights(1), synth.refinement_weights(0)
    pipe = FramePipeline(wf, wg, ws, wr, batch=B, device="cuda", boxes_per_frame=P, precision="fp32", flow_precision="fp32")
    clip = synth.clip_frames(0, B + 1, H, W).cuda()
    boxes = synth.clip_boxes(0, B, P, H, W).cuda()
    fa, fb = clip[:B].contiguous(), clip[1:B + 1].contiguous()
    pipe.step(fa, fb, boxes)                      # builds the plans (table look-ups, graph capture)
    r = pipe.step(fa, fb, boxes)                  # the replayed step bench.py times
    torch.cuda.synchronize()
    assert pipe.concurrent and pipe.refine_group == 8 and pipe.n_refine_lanes == 2
    info = ops.tune_info()
    assert info["table_sha256_16"] is not None and info["signatures_explored_by_time"] == 0
    rec = {"frames": list(FRAMES), "tune": info,
           "families": {"flow": _families(pipe.flow.plan.descs), "proposal": _families(pipe.prop_g.plan.descs),
                        "refinement": _families(pipe.refine.plan(P, H, W, False, 0, frames=8).descs)}}
    # the batch-16 choices this test exists for
    assert rec["families"]["proposal"].get("wino4x4", 0) >= 26 and rec["families"]["proposal"].get("stream", 0) > 0
    fa_h, fb_h, boxes_h = fa.cpu().numpy(), fb.cpu().numpy(), boxes.cpu().numpy()

    # ---- flow (script_pwc_multi.py:33-70 restated: cv2 resizes + PWCDCNet + rescale) ----
    rec["flow"] = {}
    for i in FRAMES:
        x0, h_, w_ = CR.flow_preprocess(fa_h[i], fb_h[i])
        with torch.no_grad():
            f2 = O.pwc_forward(wf, torch.from_numpy(x0))[0].numpy()
        ref = CR.flow_postprocess(f2, H, W, h_, w_)
        got = r["flow"][i].cpu().numpy()
        err, scale = float(np.abs(got - ref).max()), max(1.0, float(np.abs(ref).max()))
        rec["flow"][i] = {"max_abs_err_px": err, "max_abs_flow": float(np.abs(ref).max())}
        assert err < 1e-3 * scale, (i, err)

    # ---- both proposal nets (general = seed 0, specific = seed 1) ----
    nh, nw = PO.custom_resize_shape(H, W)
    rec["proposal"] = {}
    for tag, w, stage in (("general", wg, pipe.prop_g), ("specific", ws, pipe.prop_s)):
        p = stage.plan
        for i in FRAMES:
            bgr = np.ascontiguousarray(fa_h[i][:, :, ::-1])
            (fb_, fp_, fl_, fi_), inter = PO.model_forward(w, CR.resize_linear_u8(bgr, nw, nh), intermediates=True)
            n = int(p.roi_count[i].item())
            fm = p.featuremap.images(i, 1).torch().cpu()
            e_fm = float((fm - inter["featuremap"]).abs().max()) / max(1.0, float(inter["featuremap"].abs().max()))
            rpn = p.rpn_out.buf[i].cpu().numpy()
            fh, fw = rpn.shape[:2]
            lab, box = rpn[:, :, :15], rpn[:, :, 15:75].reshape(fh, fw, 15, 4)
            e_rpn = float(np.abs(lab - inter["rpn_logits"].numpy()).max()) / max(1.0, float(inter["rpn_logits"].abs().max()))
            assert e_fm < 1e-3 and e_rpn < 2e-3, (tag, i, e_fm, e_rpn)
            # strict: the fused kernel returns exactly what model.py:169-217 selects from the GPU's own logits / deltas
            dec = PO.decode_bbox_target(box, PO.all_anchors(fh, fw)).reshape(-1, 4)
            pb, ps_, pidx = PO.generate_rpn_proposals(dec, lab.reshape(-1), nh, nw)
            idx = p.roi_idx[i, :n].cpu().numpy()
            assert n == len(pidx) and np.array_equal(idx, pidx.astype(np.int32)), (tag, i)
            assert np.array_equal(p.roi_scores[i, :n].cpu().numpy(), ps_)
            # strict: the inference tail (train.py:275-295, model.py:438-491) on the GPU's own head outputs and RoIs
            head = p.head.buf.view(p.b, 100, -1)[i].cpu().numpy()
            rois = p.rois[i, :n].cpu().numpy()
            tb, tp, tl, ti = PO.fastrcnn_tail(head[:n, :2], head[:n, 2:6].reshape(n, 1, 4), rois, nh, nw)
            out = stage.net.outputs(p, i)
            assert np.array_equal(out[6], ti), (tag, i)
            assert len(out[0]) == len(tb) and (len(tb) == 0 or (np.abs(out[0] - tb).max() < 1e-2 and np.abs(out[1] - tp).max() < 1e-4))
            # across the two nets: all 100 indices shared (measured), the same position for >= 98
            common = len(np.intersect1d(idx, inter["proposal_idx"]))
            same = int(np.sum(idx[:min(n, len(inter["proposal_idx"]))] == inter["proposal_idx"][:n]))
            rec["proposal"][f"{tag}/{i}"] = {"featuremap_rel_err": e_fm, "rpn_logits_rel_err": e_rpn, "rois": n,
                                            "rpn_indices_shared_with_cpu_net": common, "rpn_indices_same_position": same,
                                            "detections": int(len(ti))}
            assert common == len(inter["proposal_idx"]) == 100, (tag, i, common)

    # ---- refinement: frame i sits in group i // 8, which ran on lane (i // 8) % 2 ----
    rec["refinement"] = {}
    for i in FRAMES:
        lane, g = (i // 8) % pipe.n_refine_lanes, i % 8
        rp = pipe.refine.plan(P, H, W, False, lane, frames=8)
        lg_all = rp.logits.torch().cpu()                          # [8 * P, 2, 97, 97]
        for b in BOXES:
            x, crop = RO.make_input(fa_h[i], boxes_h[i, b])
            with torch.no_grad():
                lg = RO.deeplab_logits(wr, x)
            glg = lg_all[g * P + b:g * P + b + 1]
            e_lg = float((glg - lg).abs().max()) / max(1.0, float(lg.abs().max()))
            rm, rpost = RO.output_layer(lg, crop, H, W)
            gm = r["masks"][i, b].cpu().numpy()
            d = gm != rm
            e_conf = abs(float(r["conf"][i, b]) - float(RO.conf_score(rm, rpost)))
            rec["refinement"][f"{i}/{b}"] = {"mask_logit_rel_err": e_lg, "conf_abs_err": e_conf, "mask_pixels_differ": int(d.sum())}
            assert e_lg < 1e-3 and e_conf < 1e-3, (i, b, e_lg, e_conf)
            assert not d.any() or np.abs(rpost[d] - 0.5).max() < 2e-3, (i, b, int(d.sum()))

    out = os.path.join(ROOT, "gpurun_out", "error_budget")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bench_object_b16.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("\nbench object (B = 16, shipped table) vs oracle:", json.dumps({k: rec[k] for k in ("flow", "proposal", "refinement")}))
