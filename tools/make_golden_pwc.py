#!/usr/bin/env python
"""Generate tests/golden/pwc_*.npz by IMPORTING the reference PWCNet.py (build container only).

The reference never travels to the GPU box; only the vectors written here do.  Recipe
(SURVEY.md 8c): stub ``correlation_package`` with a pure-torch correlation (the reference's CPU
entry points are stubs, corr.c:3-16), make ``Tensor.cuda`` the identity (PWCNet.py:166 calls it
unconditionally), and pin ``grid_sample`` to torch-0.2 semantics (bilinear/zeros/align_corners=True).

Weights: ``oracle.pwc_oracle.synth_state_dict`` loaded through the reference's own
``load_state_dict`` (so names/shapes are checked by the reference itself).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/code/optical_flow_net-PWC-Net"

from oracle import pwc_oracle as O  # noqa: E402


class _Corr(nn.Module):
    """Pure-torch stand-in with the constructor signature of modules/corr.py:4-23."""

    def __init__(self, pad_size=None, kernel_size=None, max_displacement=None, stride1=None,
                 stride2=None, corr_multiply=None):
        super().__init__()
        assert (kernel_size, stride1, stride2, corr_multiply) == (1, 1, 1, 1)
        assert pad_size == max_displacement
        self.md = max_displacement

    def forward(self, a, b):
        md = self.md
        n, c, h, w = a.shape
        bp = F.pad(b, (md, md, md, md))
        outs = []
        for dy in range(2 * md + 1):
            for dx in range(2 * md + 1):
                outs.append((a * bp[:, :, dy:dy + h, dx:dx + w]).mean(1, keepdim=True))
        return torch.cat(outs, 1)


def import_reference():
    pkg = types.ModuleType("correlation_package")
    mods = types.ModuleType("correlation_package.modules")
    corr = types.ModuleType("correlation_package.modules.corr")
    corr.Correlation = _Corr
    pkg.modules, mods.corr = mods, corr
    sys.modules.update({"correlation_package": pkg, "correlation_package.modules": mods,
                        "correlation_package.modules.corr": corr})
    torch.Tensor.cuda = lambda self, *a, **k: self
    _gs = F.grid_sample
    nn.functional.grid_sample = lambda inp, grid, *a, **k: _gs(
        inp, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    sys.path.insert(0, REF)
    import models  # the reference package
    return models


def main():
    models = import_reference()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    torch.set_grad_enabled(False)
    for tag, (h, w), wseed, fseed, shift in (
            ("64x64", (64, 64), 0, 11, (0.75, -0.5)),
            ("128x192", (128, 192), 1, 12, (1.5, -0.75))):
        net = models.pwc_dc_net(None).eval()
        sd = O.synth_state_dict(wseed)
        missing = net.load_state_dict(sd, strict=True)
        x = O.synth_frame_pair(h, w, seed=fseed, shift=shift)

        taps = {}

        def hook(name):
            def f(mod, inp, out):
                taps.setdefault(name, []).append(out.detach().clone())
            return f
        for lname in ("predict_flow6", "predict_flow5", "predict_flow4", "predict_flow3",
                      "predict_flow2", "conv6b", "conv2b", "corr"):
            getattr(net, lname).register_forward_hook(hook(lname))
        flow2 = net(x)
        arrs = {
            "wseed": np.int64(wseed), "fseed": np.int64(fseed), "shift": np.float32(shift),
            "x": x.numpy().astype(np.float16),      # in [0,1]; re-synthesised in tests, kept as a check
            "flow2": flow2.numpy(),
        }
        for lvl in (6, 5, 4, 3, 2):
            arrs[f"flow_l{lvl}"] = taps[f"predict_flow{lvl}"][0].numpy()
        arrs["c16"] = taps["conv6b"][0].numpy()
        arrs["c26"] = taps["conv6b"][1].numpy()
        arrs["c12_sub"] = taps["conv2b"][0][:, :, ::4, ::4].numpy()
        # raw (pre-LeakyReLU) cost volumes, levels 6..2, subsampled for size
        for lvl, cv in zip((6, 5, 4, 3, 2), taps["corr"]):
            s = max(1, cv.shape[-1] // 16)
            arrs[f"corr{lvl}_sub"] = cv[:, :, ::s, ::s].numpy()
        path = os.path.join(out_dir, f"pwc_{tag}.npz")
        np.savez_compressed(path, **arrs)
        print(path, os.path.getsize(path), "bytes; |flow2| max", float(flow2.abs().max()))


if __name__ == "__main__":
    main()
