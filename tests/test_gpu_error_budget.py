"""The fp32 error budget at FULL depth and full size, as recorded numbers: north_star allows 1e-3 on mask
logits and asks for bit-exact proposal indices; the conv kernels of one layer differ in rounding (implicit GEMM ~1e-6 of the
output scale, Winograd F(2x2,3x3) ~1e-6..1e-5, F(4x4,3x3) ~1e-5), and 74 layers of a pipeline step run F(4x4,3x3).  Each test
runs one net three times -- every layer on the implicit GEMM, the shipped configuration table, F(4x4,3x3) FORCED on every
layer that can run it -- against the CPU oracle and writes the measured distances to gpurun_out/error_budget/<net>.json (and
the test log); the assertions are the 1e-3 bars.  RPN indices: the strict check (GPU logits -> oracle selection == kernel
indices) lives in tests/test_gpu_fullsize.py; here the number of the 100 indices that differ from the CPU NET's is recorded."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import proposal_oracle as PO  # noqa: E402
from oracle import pwc_oracle as O  # noqa: E402
from oracle import refinement_oracle as RO  # noqa: E402

# (name, environment, MFMA arithmetic of the dense convs, held to the fp32 bars?)
MODES = [("implicit_gemm_only", {"PREMVOS_FORCE_KERNEL": "igemm"}, "fp32", True), ("shipped_table", {}, "fp32", True),
         ("f4x4_forced", {"PREMVOS_FORCE_KERNEL": "wino4"}, "fp32", True),
         # the optional bf16-MFMA modes at FULL depth (configs[2] / [4] name bf16): split-bf16 (hi.hi + hi.lo +
         # lo.hi, fp32 accumulate) must clear the same bars as fp32; plain bf16 is recorded, not asserted (it does not)
         ("bf16x3", {}, "bf16x3", True), ("bf16", {}, "bf16", False)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(net, rows):
    out = os.path.join(ROOT, "gpurun_out", "error_budget")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, net + ".json"), "w") as f:
        json.dump(rows, f, indent=1)
    print(f"\nerror budget, {net}:")
    for mode, r in rows.items():
        print("  %-20s %s" % (mode, "  ".join(f"{k}={v:.3g}" if isinstance(v, float) else f"{k}={v}" for k, v in r.items())))


def _families(descs):
    fam = {}
    for d in descs:
        k = {1: "direct", 2: "wino2x2_slab", 3: "wino2x2_fused", 4: "wino4x4", 5: "stream", 6: "bf16x3_s8"}.get(d.tile_hint, "igemm")
        fam[k] = fam.get(k, 0) + 1
    return fam


def test_flow_error_budget(monkeypatch):
    from premvos_amd.flow import pwc_dc_net
    sd = O.synth_state_dict(0)
    x = O.synth_frame_pair(512, 896, seed=7, shift=(2.5, -1.25))
    with torch.no_grad():
        ref = O.pwc_forward(sd, x)
    rows = {}
    for mode, env, prec, bars in MODES:
        for k in ("PREMVOS_FORCE_KERNEL",):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = pwc_dc_net(None, precision=prec).cuda().eval()
        net.load_state_dict(sd)
        got = net(x.cuda()).cpu()
        err = float((got - ref).abs().max())
        rows[mode] = {"flow_max_abs_err_px": err, "flow_max_abs": float(ref.abs().max()),
                      "rel_to_bar": err / (1e-3 * max(1.0, float(ref.abs().max()))), "layers": _families(net.plan(1, 512, 896).descs)}
        assert not bars or err < 1e-3 * max(1.0, float(ref.abs().max())), (mode, err)
    _record("flow_512x896", rows)
    assert rows["f4x4_forced"]["layers"].get("wino4x4", 0) >= rows["shipped_table"]["layers"].get("wino4x4", 0) > 0
    assert rows["implicit_gemm_only"]["layers"].keys() <= {"igemm", "direct"}
    assert rows["bf16"]["flow_max_abs_err_px"] > 10 * rows["bf16x3"]["flow_max_abs_err_px"]        # plain bf16 is a different accuracy class


def test_proposal_error_budget(monkeypatch):
    from oracle import cv_resize_oracle as CR
    from premvos_amd.proposal import OfflinePredictor, ProposalNet, detect_one_image
    w = PO.synth_weights(3)
    img = np.random.default_rng(3).integers(0, 256, (480, 854, 3), dtype=np.uint8)
    img = (img // 32 * 32 + np.linspace(0, 31, 854, dtype=np.uint8)[None, :, None]).astype(np.uint8)
    nh, nw = PO.custom_resize_shape(480, 854)
    (fb, fp, fl, fi), inter = PO.model_forward(w, CR.resize_linear_u8(img, nw, nh), intermediates=True)
    rows = {}
    for mode, env, prec, bars in MODES:
        monkeypatch.delenv("PREMVOS_FORCE_KERNEL", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = ProposalNet(w, precision=prec)
        detect_one_image(img, OfflinePredictor(net))
        p = net.plan(1, nh, nw)
        n = int(p.roi_count.item())
        fm = p.featuremap.torch().cpu()
        e_fm = float((fm - inter["featuremap"]).abs().max()) / max(1.0, float(inter["featuremap"].abs().max()))
        rpn = p.rpn_out.buf[0].cpu().numpy()
        lab = rpn[:, :, :15]
        e_rpn = float(np.abs(lab - inter["rpn_logits"].numpy()).max()) / max(1.0, float(inter["rpn_logits"].abs().max()))
        idx = p.roi_idx[0, :n].cpu().numpy()
        common = len(np.intersect1d(idx, inter["proposal_idx"]))
        same_order = int(np.sum(idx[:min(n, len(inter["proposal_idx"]))] == inter["proposal_idx"][:n]))
        rows[mode] = {"featuremap_rel_err": e_fm, "rpn_logits_rel_err": e_rpn, "rois": n,
                      "rpn_indices_shared_with_cpu_net": common, "rpn_indices_same_position": same_order,
                      "layers": _families(p.descs), "pointwise_layers_on_the_split_bf16_kernel": p.split_layers}
        assert not bars or (e_fm < 1e-3 and e_rpn < 2e-3), (mode, e_fm, e_rpn)
        # asserted as measured (VERDICT r03 next #1c): every fp32-class mode shares all 100 indices with the CPU net, 98 of them at
        # the same rank (the two swapped ranks are near-ties of the CPU net itself, the same two in every mode)
        assert not bars or (common == 100 and same_order >= 98), (mode, common, same_order)
    _record("proposal_749x1333", rows)
    assert rows["bf16x3"]["pointwise_layers_on_the_split_bf16_kernel"] > 0 == rows["shipped_table"]["pointwise_layers_on_the_split_bf16_kernel"]
    assert rows["f4x4_forced"]["layers"].get("wino4x4", 0) >= rows["shipped_table"]["layers"].get("wino4x4", 0)


def test_refinement_error_budget(monkeypatch):
    from premvos_amd.refinement import RefinementNet
    w = RO.synth_weights(4, 16)
    H, W = 480, 854
    boxes = [[100.0, 200.0, 300.0, 500.0], [0.0, 0.0, 480.0, 854.0], [400.2, 800.7, 470.0, 850.0]]
    img = (np.random.default_rng(H).random((H, W, 3)) * 255).astype(np.uint8)
    refs = []
    for b in boxes:
        x, crop = RO.make_input(img, b)
        with torch.no_grad():
            lg = RO.deeplab_logits(w, x)
        refs.append((lg, crop) + RO.output_layer(lg, crop, H, W))
    rows = {}
    for mode, env, prec, bars in MODES:
        monkeypatch.delenv("PREMVOS_FORCE_KERNEL", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = RefinementNet(w, 16, precision=prec)
        p = net.refine(torch.from_numpy(img).cuda(), torch.tensor(boxes).cuda(), max_boxes=4, with_posterior=True)
        e_lg = e_post = 0.0
        flips = 0
        for i, (lg, crop, rm, rp) in enumerate(refs):
            glg = p.logits.torch().cpu()[i:i + 1]
            e_lg = max(e_lg, float((glg - lg).abs().max()) / max(1.0, float(lg.abs().max())))
            e_post = max(e_post, float(np.abs(p.posterior[i].cpu().numpy() - rp).max()))
            flips += int((p.mask[i].cpu().numpy() != rm).sum())
        rows[mode] = {"mask_logit_rel_err": e_lg, "posterior_max_abs_err": e_post, "mask_pixels_flipped_of": f"{flips}/{3 * H * W}",
                      "layers": _families(net.plan(4, H, W, True).descs)}
        assert not bars or (e_lg < 1e-3 and e_post < 1e-3), (mode, e_lg, e_post)
    _record("refinement_385x385", rows)


def test_configs4_1080p_error_budget(monkeypatch):
    """configs[4] in the arithmetic it names (VERDICT r03 next #1b): a 1080x1920 frame -- flow at 1088x1920, proposal_net at
    750x1333 (the other rounding of CustomResize), refinement crops of a 1080p frame -- at FULL depth in fp32 (shipped table),
    bf16x3 (split-fp32 on the bf16 MFMA pipe; held to the fp32 bars) and plain bf16 (recorded).  Flow stays fp32 in the mixed
    modes bench.py offers; its bf16x3 row is recorded and held to the bar as well."""
    from oracle import cv_resize_oracle as CR
    from premvos_amd.flow import pwc_dc_net
    from premvos_amd.proposal import OfflinePredictor, ProposalNet, detect_one_image
    from premvos_amd.refinement import RefinementNet
    modes = [m for m in MODES if m[0] in ("shipped_table", "bf16x3", "bf16")]
    rows = {m[0]: {} for m in modes}
    # ---- flow, 1088x1920 ----
    sd = O.synth_state_dict(0)
    x = O.synth_frame_pair(1088, 1920, seed=7, shift=(2.5, -1.25))
    with torch.no_grad():
        ref = O.pwc_forward(sd, x)
    for mode, env, prec, bars in modes:
        monkeypatch.delenv("PREMVOS_FORCE_KERNEL", raising=False)
        net = pwc_dc_net(None, precision=prec).cuda().eval()
        net.load_state_dict(sd)
        err = float((net(x.cuda()).cpu() - ref).abs().max())
        rows[mode]["flow_1088x1920"] = {"flow_max_abs_err_px": err, "flow_max_abs": float(ref.abs().max())}
        assert not bars or err < 1e-3 * max(1.0, float(ref.abs().max())), (mode, err)
        del net
        torch.cuda.empty_cache()
    # ---- proposal_net, 1080x1920 -> 750x1333 ----
    w = PO.synth_weights(5)
    rng = np.random.default_rng(5)
    img = (rng.integers(0, 8, (1080, 1920, 3)) * 32 + np.linspace(0, 31, 1920, dtype=np.uint8)[None, :, None]).astype(np.uint8)
    nh, nw = PO.custom_resize_shape(1080, 1920)
    assert (nh, nw) == (750, 1333)
    (fb, fp, fl, fi), inter = PO.model_forward(w, CR.resize_linear_u8(img, nw, nh), intermediates=True)
    for mode, env, prec, bars in modes:
        net = ProposalNet(w, precision=prec)
        detect_one_image(img, OfflinePredictor(net))
        p = net.plan(1, nh, nw)
        n = int(p.roi_count.item())
        fm = p.featuremap.torch().cpu()
        e_fm = float((fm - inter["featuremap"]).abs().max()) / max(1.0, float(inter["featuremap"].abs().max()))
        rpn = p.rpn_out.buf[0].cpu().numpy()
        fh, fw = rpn.shape[:2]
        lab = rpn[:, :, :15]
        e_rpn = float(np.abs(lab - inter["rpn_logits"].numpy()).max()) / max(1.0, float(inter["rpn_logits"].abs().max()))
        idx = p.roi_idx[0, :n].cpu().numpy()
        common = len(np.intersect1d(idx, inter["proposal_idx"]))
        same = int(np.sum(idx[:min(n, len(inter["proposal_idx"]))] == inter["proposal_idx"][:n]))
        # the kernel's own selection is strict in every mode: model.py:169-217 on THIS mode's logits
        dec = PO.decode_bbox_target(rpn[:, :, 15:75].reshape(fh, fw, 15, 4), PO.all_anchors(fh, fw)).reshape(-1, 4)
        pb, ps, pidx = PO.generate_rpn_proposals(dec, lab.reshape(-1), nh, nw)
        assert n == len(pidx) and np.array_equal(idx, pidx.astype(np.int32)), mode
        rows[mode]["proposal_750x1333"] = {"featuremap_rel_err": e_fm, "rpn_logits_rel_err": e_rpn, "rois": n,
                                           "rpn_indices_shared_with_cpu_net": common, "rpn_indices_same_position": same,
                                           "layers": _families(p.descs)}
        assert not bars or (e_fm < 1e-3 and e_rpn < 2e-3 and common >= 95), (mode, e_fm, e_rpn, common)     # (measured at this shape: 97 shared in fp32 -- near-ties of the CPU net; the strict check above is the index contract)
        del net, p
        torch.cuda.empty_cache()
    # ---- refinement_net on a 1080x1920 frame ----
    rw = RO.synth_weights(4, 16)
    H, W = 1080, 1920
    boxes = [[500.5, 900.5, 900.0, 1500.0], [0.0, 1800.0, 60.0, 1920.0], [100.0, 100.0, 1000.0, 1700.0]]
    frame = (np.random.default_rng(H).random((H, W, 3)) * 255).astype(np.uint8)
    refs = []
    for b in boxes:
        xin, crop = RO.make_input(frame, b)
        with torch.no_grad():
            lg = RO.deeplab_logits(rw, xin)
        refs.append((lg, crop) + RO.output_layer(lg, crop, H, W))
    for mode, env, prec, bars in modes:
        net = RefinementNet(rw, 16, precision=prec)
        p = net.refine(torch.from_numpy(frame).cuda(), torch.tensor(boxes).cuda(), max_boxes=4, with_posterior=True)
        e_lg = e_post = 0.0
        flips = 0
        for i, (lg, crop, rm, rp) in enumerate(refs):
            glg = p.logits.torch().cpu()[i:i + 1]
            e_lg = max(e_lg, float((glg - lg).abs().max()) / max(1.0, float(lg.abs().max())))
            e_post = max(e_post, float(np.abs(p.posterior[i].cpu().numpy() - rp).max()))
            flips += int((p.mask[i].cpu().numpy() != rm).sum())
        rows[mode]["refinement_1080p_frame"] = {"mask_logit_rel_err": e_lg, "posterior_max_abs_err": e_post,
                                                "mask_pixels_flipped_of": f"{flips}/{3 * H * W}"}
        assert not bars or (e_lg < 1e-3 and e_post < 1e-3), (mode, e_lg, e_post)
        del net, p
        torch.cuda.empty_cache()
    _record("configs4_1080p", rows)
