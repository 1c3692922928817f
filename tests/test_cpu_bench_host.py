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
    assert a.supplementary == "mixed-bf16x3,1080p" and a.gpus == 1 and a.frames == 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--supplementary", "none", "--frames", "40"])
    a = bench.parse()
    assert [m for m in a.supplementary.split(",") if m in ("mixed-bf16x3", "1080p")] == [] and a.frames == 40
