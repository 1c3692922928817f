"""CPU suite (-m "not gpu"): oracle vs the reference-generated golden vectors, host logic, C-ABI surface."""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cv_resize_oracle as R
from oracle import pwc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tag", ["64x64", "128x192"])
def test_pwc_oracle_matches_reference_golden(golden_dir, tag):
    """The vectors were produced by importing the reference PWCNet.py (tools/make_golden_pwc.py)."""
    g = np.load(os.path.join(golden_dir, f"pwc_{tag}.npz"))
    h, w = map(int, tag.split("x"))
    x = O.synth_frame_pair(h, w, seed=int(g["fseed"]), shift=tuple(float(s) for s in g["shift"]))
    assert np.abs(x.numpy() - g["x"].astype(np.float32)).max() < 5e-4      # stored as f16
    with torch.no_grad():
        f, inter = O.pwc_forward(O.synth_state_dict(int(g["wseed"])), x, intermediates=True)
    assert np.abs(f.numpy() - g["flow2"]).max() < 1e-5
    for lvl in (6, 5, 4, 3, 2):
        assert np.abs(inter[f"flow{lvl}"].numpy() - g[f"flow_l{lvl}"]).max() < 1e-5
    assert np.abs(inter["c16"].numpy() - g["c16"]).max() < 1e-5
    assert np.abs(inter["c26"].numpy() - g["c26"]).max() < 1e-5
    assert np.abs(inter["c12"][:, :, ::4, ::4].numpy() - g["c12_sub"]).max() < 1e-5
    # raw cost volumes (the reference tap is before the LeakyReLU)
    for lvl in (6, 5, 4, 3, 2):
        cv = inter[f"corr{lvl}"]
        raw = torch.where(cv < 0, cv / 0.1, cv)
        s = max(1, cv.shape[-1] // 16)
        assert np.abs(raw[:, :, ::s, ::s].numpy() - g[f"corr{lvl}_sub"]).max() < 1e-4


def test_state_dict_names_match_reference_layout():
    shapes = O.param_shapes()
    assert len(shapes) == 2 * (18 + 5 * 5 + 5 + 5 + 4 + 6 + 1)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 9374340   # the reference module reports 9 374 340 params
    assert shapes["conv6_0.0.weight"] == (128, 81, 3, 3)
    assert shapes["predict_flow2.weight"] == (2, 565, 3, 3)
    assert shapes["upfeat5.weight"] == (661, 2, 4, 4)
    assert shapes["dc_conv1.0.weight"] == (128, 565, 3, 3)


def test_correlation_known_answer_from_reference_test():
    """correlation-pytorch/test/test.py:76-77 and :81."""
    a = np.array([[1, 2], [3, 4]], np.float32).reshape(1, 1, 2, 2)
    b = np.array([[5, 6], [7, 8]], np.float32).reshape(1, 1, 2, 2)
    assert np.array_equal(O.correlation_np(a, b, 0, 1, 0, 1, 1).reshape(2, 2), [[5, 12], [21, 32]])
    assert O.correlation_np(a, b, 1, 1, 1, 1, 1).shape == (1, 9, 2, 2)
    # test.ipynb: Correlation(40,1,40,1,1,1) on 128x100x100 -> 81x81 displacement channels
    assert O.correlation_np(np.zeros((1, 2, 10, 10), np.float32), np.zeros((1, 2, 10, 10), np.float32),
                            4, 1, 4, 1, 1).shape == (1, 81, 10, 10)


def test_correlation_fast_path_equals_general():
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn((2, 12, 9, 11), generator=g), torch.randn((2, 12, 9, 11), generator=g)
    assert np.abs(O.correlation_torch(a, b).numpy() - O.correlation_np(a.numpy(), b.numpy())).max() < 1e-6


def test_warp_identity_and_out_of_bounds():
    x = torch.randn((1, 3, 6, 7))
    assert torch.allclose(O.warp(x, torch.zeros((1, 2, 6, 7))), x, atol=1e-6)
    far = torch.full((1, 2, 6, 7), 100.0)
    assert O.warp(x, far).abs().max().item() == 0.0
    one = torch.zeros((1, 2, 6, 7))
    one[:, 0] = 1.0                                         # sample from x+1
    out = O.warp(x, one)
    assert torch.allclose(out[..., :-1], x[..., 1:], atol=1e-6) and out[..., -1].abs().max().item() == 0.0


def test_cv_resize_restatement_sanity():
    """No cv2 here (parity unpinned): the restatement must at least agree with half-pixel bilinear."""
    im = np.random.default_rng(0).integers(0, 256, (48, 85, 3), dtype=np.uint8)
    r = R.resize_linear_u8(im, 128, 64)
    t = torch.from_numpy(im).permute(2, 0, 1)[None].float()
    f = F.interpolate(t, size=(64, 128), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(r.astype(np.float32) - f).max() <= 1.0
    assert np.array_equal(R.resize_linear_u8(im, 85, 48), im)
    g = np.random.default_rng(1).standard_normal((16, 28)).astype(np.float32)
    ff = F.interpolate(torch.from_numpy(g)[None, None], size=(48, 85), mode="bilinear", align_corners=False)
    assert np.abs(R.resize_linear_f32(g, 85, 48) - ff[0, 0].numpy()).max() < 1e-4
    const = np.full((10, 10, 3), 77, np.uint8)
    assert (R.resize_linear_u8(const, 64, 64) == 77).all()


def test_pack_deconv_is_a_3x3_conv_plus_pixel_shuffle():
    from premvos_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn((2, 5, 6, 7), generator=g)
    w = torch.randn((5, 3, 4, 4), generator=g)
    b = torch.randn((3,), generator=g)
    ref = F.conv_transpose2d(x, w, b, stride=2, padding=1)
    pk = ops.pack_deconv4x4s2(w, b, device="cpu")
    assert pk.cout == 12 and pk.cout_ps == 3 and pk.kh == 3 and pk.cin_pad == 8 and pk.k_pad % 16 == 0
    w3 = pk.wgt[:12, :9 * 8].reshape(12, 9, 8)[:, :, :5].reshape(12, 3, 3, 5).permute(0, 3, 1, 2)
    y = F.conv2d(x, w3, pk.bias[:12], padding=1)                  # [2,12,6,7], phase-major channels
    y = y.view(2, 2, 2, 3, 6, 7).permute(0, 3, 4, 1, 5, 2).reshape(2, 3, 12, 14)
    assert torch.allclose(y, ref, atol=1e-5)


def test_pack_conv_layout():
    from premvos_amd import ops
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).view(2, 3, 3, 3)
    pk = ops.pack_conv(w, torch.tensor([1.0, 2.0]), device="cpu")
    assert (pk.cin_pad, pk.k_pad, pk.cout_pad) == (4, 48, 32)
    k = (1 * 3 + 2) * 4 + 1                                       # kh=1, kw=2, c=1
    assert pk.wgt[1, k].item() == w[1, 1, 1, 2].item()
    assert pk.wgt[:, 36:].abs().max().item() == 0 and pk.wgt[2:].abs().max().item() == 0
    assert pk.wgt[0, 3].item() == 0                               # channel pad lane
    sc = ops.pack_conv(w, None, device="cpu", scale=torch.tensor([2.0, 0.5]))
    assert sc.wgt[1, k].item() == 0.5 * w[1, 1, 1, 2].item() and sc.bias is None


def test_flo_writer_layout(tmp_path):
    """script_pwc_multi.py:16-31 byte layout; reader = MergeTrack/merge_functions.py:197-207."""
    from premvos_amd.flow.driver import readFlowFile, writeFlowFile
    uv = np.random.default_rng(0).standard_normal((5, 7, 2)).astype(np.float32)
    fn = str(tmp_path / "a.flo")
    writeFlowFile(fn, uv)
    raw = open(fn, "rb").read()
    assert len(raw) == 12 + 5 * 7 * 2 * 4
    assert np.frombuffer(raw[:4], np.float32)[0] == np.float32(202021.25)
    assert tuple(np.frombuffer(raw[4:12], np.int32)) == (7, 5)
    assert np.array_equal(readFlowFile(fn), uv)
    with pytest.raises(ValueError):
        writeFlowFile(fn, np.zeros((5, 7, 3), np.float32))


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    """No compute calls (no GPU here): the .so must load and export what include/*.h declares."""
    from premvos_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "premvos_hip.h")).read()
    declared = set(re.findall(r"\b(premvos_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in premvos_hip.h but not exported"
    assert declared - {"premvos_last_error", "premvos_abi_version", "premvos_refine_output_workspace_bytes",
                       "premvos_conv2d_workspace_bytes", "premvos_crc32c_host", "premvos_rle_workspace_bytes",
                       "premvos_rle_counts_to_string_host", "premvos_rle_strings_host", "premvos_write_frame_files_host", "premvos_format_floats_host", "premvos_jpeg_workspace_bytes"} == set(
        _lib.SIGNATURES)
    assert lib.premvos_abi_version() >= 1
    # argument validation happens before any HIP call, so it is testable without a GPU
    import ctypes as C
    d = _lib.ConvDesc()
    assert lib.premvos_conv2d_f32(C.byref(d), None) == -1
    assert b"null" in lib.premvos_last_error()
    assert lib.premvos_corr_fwd_f32(None, 4, None, 4, None, 81, 1, 2, 2, 4, 4, 0.1, 0, None) == -1


def test_product_path_refuses_to_run_without_gpu():
    from premvos_amd import _lib
    from premvos_amd.flow import pwc_dc_net
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = pwc_dc_net(None)
    with pytest.raises(_lib.PremvosError):
        net.load_state_dict(O.synth_state_dict(0))
    with pytest.raises(_lib.PremvosError):
        net(torch.zeros((1, 6, 64, 64)))
    # every other entry of the product path: no CPU fallback anywhere
    from premvos_amd import mergetrack
    from premvos_amd.proposal import ProposalNet
    from premvos_amd.refinement import RefinementNet
    from premvos_amd.reid import ReIDNet
    for ctor in (lambda: ProposalNet({}, (1, 1, 1, 1)), lambda: RefinementNet({}, 1), lambda: ReIDNet({}),
                 lambda: mergetrack.warp_masks(np.zeros((1, 4, 4), np.uint8), np.zeros((4, 4, 2), np.float32)),
                 lambda: mergetrack.mask_iou(np.zeros((1, 4, 4), np.uint8), np.zeros((1, 4, 4), np.uint8)),
                 lambda: mergetrack.encode_masks(np.zeros((1, 4, 4), np.uint8))):
        with pytest.raises(_lib.PremvosError):
            ctor()


def test_product_path_never_imports_the_oracle():
    """A product path that routes through oracle/ voids every parity claim."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "premvos_amd")):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
                    bad.append(os.path.join(dirpath, fn))
    assert not bad, bad


def test_winograd_filter_transforms_reproduce_the_convolution():
    """Host logic of the two Winograd paths (ops.pack_winograd: F(2x2,3x3), ops.pack_winograd4: F(4x4,3x3)): the packed filter
    transforms U = G g G^T, multiplied with B^T d B and folded with A^T . A in float64, give the direct 3x3 convolution (the
    matrices the kernels of csrc/conv_wino_f32.hip / conv_wino4_f32.hip implement)."""
    from premvos_amd import ops
    g = torch.Generator().manual_seed(3)
    cin, cout, h, w = 20, 12, 9, 7
    x = torch.randn((1, cin, h, w), generator=g, dtype=torch.float64)
    wt = torch.randn((cout, cin, 3, 3), generator=g, dtype=torch.float64) * 0.1
    ref = torch.nn.functional.conv2d(x, wt, padding=1)
    variants = {
        2: (ops.pack_winograd, torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64),
            torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)),
        4: (ops.pack_winograd4, torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                                              [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64),
            torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)),
    }
    for m, (pack, BT, AT) in variants.items():
        t = m + 2
        packed = pack(wt.float(), 20, 32, "cpu")
        assert tuple(packed.shape) == (t * t, 32, 32) and packed.dtype == torch.float32
        assert packed[:, cout:].abs().max() == 0 and packed[:, :, cin:].abs().max() == 0       # zero padded
        U = packed.double()[:, :cout, :cin].reshape(t, t, cout, cin)
        ty, tx = -(-h // m), -(-w // m)
        xp = torch.nn.functional.pad(x, (1, m * tx + 1 - w, 1, m * ty + 1 - h))
        out = torch.zeros((cout, m * ty, m * tx), dtype=torch.float64)
        for a in range(ty):
            for b in range(tx):
                d = xp[0, :, m * a:m * a + t, m * b:m * b + t]
                V = torch.einsum("ia,cab,jb->ijc", BT, d, BT)
                M = torch.einsum("ijc,ijoc->ijo", V, U)
                out[:, m * a:m * a + m, m * b:m * b + m] = torch.einsum("ai,ijo,bj->oab", AT, M, AT)
        assert (out[:, :h, :w] - ref[0]).abs().max().item() < 1e-5, m         # (fp32 rounding of the packed filters)
