"""Host-side pieces of bench.py that need no GPU: the clock / power sampler (on a fake amdsmi), the canonical-layer pick and the
argument surface of the supplementary passes."""
import importlib
import sys
import time
import types

import pytest


@pytest.fixture()
def bench():
    sys.modules.pop("bench", None)
    return importlib.import_module("bench")


def _fake_amdsmi(monkeypatch, readings, fail_after=None):
    calls = {"n": 0}
    m = types.ModuleType("amdsmi")
    m.amdsmi_init = lambda: None
    m.amdsmi_get_processor_handles = lambda: ["gpu0", "gpu1"]

    def metrics(h):
        i = calls["n"]
        calls["n"] += 1
        if fail_after is not None and i >= fail_after:
            raise RuntimeError("gone")
        return readings[min(i, len(readings) - 1)]
    m.amdsmi_get_gpu_metrics_info = metrics
    monkeypatch.setitem(sys.modules, "amdsmi", m)
    return calls


def test_box_sampler_summarises_clock_power_and_power_limited_share(bench, monkeypatch):
    rd = [{"current_gfxclks": [2300, 2250, 2280, 2270, "N/A", 65535, 2260, 2290], "current_socket_power": 1340 + i, "temperature_hotspot": 56,
           "temperature_mem": 50, "current_uclk": 2000, "ppt_residency_acc": 100 + 6 * i, "accumulation_counter": 1000 + 10 * i} for i in range(50)]
    _fake_amdsmi(monkeypatch, rd)
    with bench.BoxSampler(1, period_s=0.002) as s:
        time.sleep(0.05)
    out = s.summary()
    assert out["samples"] >= 3 and "error" not in out
    assert out["sclk_mhz_mean_of_xcds"]["mean"] == 2275.0 and out["sclk_mhz_slowest_xcd"]["min"] == 2250      # "N/A" and 65535 are not clocks
    assert 1340 <= out["socket_power_w"]["min"] <= out["socket_power_w"]["max"] <= 1390
    assert out["power_limited_share"] == 0.6 and out["temperature_hotspot_c"]["max"] == 56


def test_box_sampler_without_amdsmi_reports_an_error_and_does_not_raise(bench, monkeypatch):
    monkeypatch.setitem(sys.modules, "amdsmi", None)             # import amdsmi -> ImportError
    with bench.BoxSampler(0) as s:
        pass
    assert "error" in s.summary() and "samples" not in s.summary()
    _fake_amdsmi(monkeypatch, [{"current_gfxclk": 2100, "current_socket_power": 900}], fail_after=3)   # dies mid-run: keeps what it has
    with bench.BoxSampler(0, period_s=0.001) as s:
        time.sleep(0.03)
    out = s.summary()
    assert out["samples"] == 2 and out["sclk_mhz_mean_of_xcds"]["mean"] == 2100.0 and out["power_limited_share"] is None


def test_canonical_layer_is_the_middle_flow_pointwise_with_residual(bench):
    d = types.SimpleNamespace(cin=728, cout=728, n=160, ho=25, wo=25)
    items = [("flow", "conv:conv1a", None, 1.0, 1.0, d),
             ("refine", "conv:middle_flow/block1/unit_8/xception_module/separable_conv3_pointwise", None, 2.0 * 1e5 * 728 * 728 * 2, 1.0, d)]
    out = bench._canonical_layer(items, [0.1, 0.9])["canonical_layer"]
    assert out["us"] == 900.0 and abs(out["tflops"] - 2.0 * 1e5 * 728 * 728 / 0.9e-3 / 1e12) < 0.06 and "728 -> 728" in out["what"]
    assert bench._canonical_layer(items[:1], [0.1])["canonical_layer"] is None


def test_supplementary_modes_are_an_argument(bench, monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.supplementary == "mixed-bf16x3,1080p,1080p-mixed-bf16x3" and a.gpus == 1 and a.frames == 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--supplementary", "none", "--frames", "40"])
    a = bench.parse()
    assert [m for m in a.supplementary.split(",") if m in ("mixed-bf16x3", "1080p")] == [] and a.frames == 40


def test_box_sampler_picks_the_amdsmi_device_by_pci_address_not_by_position(bench, monkeypatch):
    """amdsmi lists every physical GPU whatever HIP_VISIBLE_DEVICES says; HIP device 0 may be amdsmi's third."""
    import torch
    _fake_amdsmi(monkeypatch, [{"current_gfxclk": 2100, "current_socket_power": 900}])
    smi = sys.modules["amdsmi"]
    smi.amdsmi_get_processor_handles = lambda: ["a", "b", "c"]
    smi.amdsmi_get_gpu_device_bdf = lambda h: {"a": "0000:05:00.0", "b": "0000:c1:00.0", "c": "0001:0a:00.0"}[h]
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: types.SimpleNamespace(pci_domain_id=1, pci_bus_id=0x0A, pci_device_id=0))
    s = bench.BoxSampler(0)
    assert s.err is None and s.h == "c" and s.matched_by == "pci 0001:0a:00.0"
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0x77, pci_device_id=0))
    s = bench.BoxSampler(0)
    assert s.err is not None and "no amdsmi device at PCI 0000:77:00" in s.err and "error" in s.summary()


def test_traffic_figure_covers_every_fp32_conv_kernel_the_plans_can_launch():
    """`roofline.traffic` is read from profiles/rNN_conv_hbm_traffic.json; round 5's file had been produced by a
    kernel list that lacked the round's new kernel.  Every __global__ of the fp32 conv sources (what premvos_conv2d_f32 dispatches
    to) must be in tools/pmc_traffic.py's list, and in the list the newest traffic file (round >= 6) was produced with."""
    import glob
    import json
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for fn in glob.glob(os.path.join(root, "premvos_amd", "csrc", "conv_*.hip")):
        src = open(fn).read()
        names |= set(re.findall(r"__global__[^;{]*?\bvoid\s+([a-z0-9_]+)\s*\(", src, flags=re.S))
    conv = {n for n in names if "bf16" not in n and "calibrate" not in n and n not in ("digest_kernel", "split8_kernel")}
    assert {"conv_igemm_f32_kernel", "conv_pwdma_f32_kernel", "wino4_gemm_kernel", "conv_stream_f32_kernel"} <= conv, conv
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import pmc_traffic
    finally:
        sys.path.pop(0)
    assert conv <= set(pmc_traffic.KERNELS), conv - set(pmc_traffic.KERNELS)
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_conv_hbm_traffic.json")))
    newest = [f for f in files if int(os.path.basename(f)[1:3]) >= 6][-1:]
    for f in newest:
        assert conv <= set(json.load(open(f))["kernels"]), (f, conv - set(json.load(open(f))["kernels"]))
