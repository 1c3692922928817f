"""bench.py's driver contract, exercised end to end on one GPU: ONE JSON line with the contract's keys, and the
multi-GPU code path (process group over RCCL, per-step gather to the merge rank, max-over-ranks timing) with world
size 1 (PREMVOS_BENCH_FORCE_DIST=1) -- the 8-GPU launch itself is the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args):
    env = dict(os.environ, **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_through_the_distributed_path():
    d = _run({"PREMVOS_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533", "RANK": "0",
              "WORLD_SIZE": "1", "LOCAL_RANK": "0"}, "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
             "--scaling", "weak", "--no-cpu-baseline", "--file-to-file", "0", "--supplementary", "none")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert abs(d["value"] - 2 * 1000.0 / d["ms_per_step"]) < 1e-2 * d["value"]          # frames / time, whole job
    assert d["value"] > 10.0                                                               # an MI355X does > 10 frames/s
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 20 < r["achieved"] < r["mfma_ceiling_measured"] <= 160
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "ONE" in d["config"]["parallelism"] and "gather" in d["config"]["parallelism"]
    assert r["traffic"] is None or "profiles/" in r["traffic_source"]
    # the line carries its own box factors: HBM copy rate, one canonical layer, clock / power over the timed region
    assert 2000 < r["hbm_ceiling_measured"] < 8000 and r["canonical_layer"]["us"] > 0 and "728 -> 728" in r["canonical_layer"]["what"]
    assert "error" in d["box"] or (d["box"]["samples"] >= 1 and 500 < d["box"]["sclk_mhz_mean_of_xcds"]["mean"] < 3000)
    sr = r["mfma_ceiling_sustained_random_data"]              # the pure MFMA loop on changing random operands, sustained: <= the burst on constants
    assert 100 < sr["tflops"] <= r["mfma_ceiling_measured"] * 1.01 and sr["seconds"] >= 1.0
    # the roofline leg's own clock sample and the peak / family fraction at that clock (`frac` itself stays priced at the nominal peak)
    assert "error" in r["leg_box"] or (r["peak_at_granted_clock"] <= 157.3 * 1.02 and
                                       abs(r["igemm_family_frac_at_granted_clock"] * r["peak_at_granted_clock"] - r["igemm_family_frac"] * 157.3) < 0.5)
    # ... and per-rank scaling diagnostics whenever a process group exists
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == 1 and pr["rank_skew_ms_per_step"] == 0.0 and pr["exchange_wait_ms_per_step"][0] >= 0.0
    assert pr["cold_start_s"][0] > 0 and d["cold_start_s"] == pr["cold_start_s"][0]
    # round 6: the whole launch priced against the same peak (every kernel of a frame, the timed region's own wall time), and a cold
    # start that no longer counts the synthesis of weights / frames on the host
    gf = d["config"]["gflop_per_frame"]
    assert abs(r["whole_step_frac"] - 2 * gf * 1e9 / (r["whole_step_ms_per_launch"] * 1e-3) / 1e12 / 157.3) < 2e-3
    assert 0.05 < r["whole_step_frac"] < r["frac"] + 0.05 and abs(r["non_conv_ms_per_launch"] - (r["whole_step_ms_per_launch"] - r["conv_ms_per_step"])) < 1e-2
    cs = d["cold_start"]
    assert abs(cs["build_pipeline_s"] + cs["first_launch_s"] - d["cold_start_s"]) < 0.03 and cs["synthetic_inputs_s"] > 0
    assert d["cold_start_s"] < 30 and d["box"].get("device_matched_by", "pci").startswith(("pci", "position"))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` outside torch.distributed spawns the ranks itself; with the gloo backend two ranks may share
    the one GPU of the test box (RCCL, the product backend, needs a device per rank)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"PREMVOS_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PREMVOS_AUTOTUNE": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch",
                        "2", "--scaling", "weak", "--no-cpu-baseline", "--no-roofline", "--file-to-file", "0"], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["frames_per_launch_per_gpu"] == 2 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 2 * 1000.0 / d["ms_per_step"]) < 1e-2 * d["value"]      # whole-job frames / max-over-ranks time


def test_bench_strong_scaling_shards_one_video_over_two_ranks():
    """The default mode: ONE synthetic video shared out by premvos_amd.parallel.plan_shards -- 7 frame pairs in chunks of 2 over
    two ranks = ranges [0,4) and [4,7): rank 0 reads frame 4 as the second image of its last pair, rank 1's last chunk is ragged
    (one frame, padded to the launch batch, counted once).  value = frames of the video / max-over-ranks time per pass."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"PREMVOS_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch",
                        "2", "--frames", "7", "--no-cpu-baseline", "--no-roofline", "--file-to-file", "0"], capture_output=True, text=True, env=env,
                       timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["frames_per_step"] == 7
    assert abs(d["value"] - 7 * 1000.0 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert d["conv_configurations"]["signatures_explored_by_time"] == 0                   # nothing was chosen by wall clock


def test_bench_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PREMVOS_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "GPU" in (r.stderr + r.stdout)


def test_bench_strong_scaling_eight_ranks_with_a_ragged_video():
    """World 8 over gloo on the one test GPU: 13 frame pairs in chunks of 2 = 7 chunks over 8 ranks -- one rank
    owns no chunk and only takes part in the gathers, the last chunk is ragged.  The line counts the 13 frames once."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"PREMVOS_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch",
                        "2", "--frames", "13", "--no-cpu-baseline", "--no-roofline", "--file-to-file", "0"], capture_output=True, text=True,
                       env=env, timeout=2400)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["frames_per_step"] == 13
    assert abs(d["value"] - 13 * 1000.0 / d["ms_per_step"]) < 1e-2 * d["value"]
    # per-rank scaling diagnostics: one entry per rank, the chunk-less rank launches nothing
    pr = d["per_rank"]
    assert all(len(pr[k]) == 8 for k in ("ms_per_step", "launches_per_step", "exchange_wait_ms_per_step", "cold_start_s"))
    assert sorted(pr["launches_per_step"]) == [0, 1, 1, 1, 1, 1, 1, 1] and sum(pr["launches_per_step"]) == 7
    assert pr["rank_skew_ms_per_step"] == round(max(pr["ms_per_step"]) - min(pr["ms_per_step"]), 3) and 0 <= pr["slowest_rank"] < 8
    assert max(pr["ms_per_step"]) <= d["ms_per_step"] * 1.001 and all(w >= 0 for w in pr["exchange_wait_ms_per_step"])
    assert d["conv_configurations"]["signatures_explored_by_time"] == 0
