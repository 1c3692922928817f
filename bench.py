#!/usr/bin/env python
"""bench.py -- the driver contract.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Metric (BASELINE.json): frames/sec of (proposal + refine + flow) on 480p DAVIS-shape frames.
A "step" = one pass of the whole per-frame hot path over a batch of B synthetic 480x854 uint8 frames that are
resident in HBM when the clock starts:
    PWC-Net flow of (t, t+1)  +  proposal_net with the general AND the specific weight set (simple_run.sh:28-42)
    +  refinement_net on P = 20 boxes per frame                                  (2.42 TFLOP/frame)
Refinement boxes are the seeded synthetic boxes SURVEY.md 8(d) prescribes (random-weight proposals are
meaningless); the two proposal passes still run completely (100 RoIs each) inside the timed region.
Each rank owns its own frames (no data-path collective); per step every rank's results (flow, bit-packed masks, conf,
boxes) go to rank 0 -- the merge rank -- in ONE RCCL gather of one fixed-size buffer (premvos_amd.parallel.ResultExchange),
inside the timed region.  Rank 0 prints ONE JSON line.

`python bench.py --gpus N` with N > 1 and no torch.distributed environment launches its own N ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one process per GPU) and fails
loudly when the node has fewer than N GPUs.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32) = vector peak
H, W = 480, 854              # DAVIS 480p
P_BOXES = 20                 # RESULTS_PER_IM (proposal_net/config.py:123); SURVEY 8(d)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PREMVOS_BENCH_BATCH", "16")),
                    help="frames per step per GPU")
    ap.add_argument("--precision", default=os.environ.get("PREMVOS_BENCH_PRECISION", "fp32"),
                    choices=["fp32", "bf16x3", "bf16", "mixed-bf16x3", "mixed-bf16"],
                    help="MFMA arithmetic of the dense convs; mixed-*: PWC-Net fp32, proposal/refinement in the bf16 mode")
    ap.add_argument("--frame", default="480p", choices=["480p", "1080p"],
                    help="480p = the metric's DAVIS shape (default); 1080p = configs[4]'s 1080x1920 frames (supplementary)")
    ap.add_argument("--scaling", default=os.environ.get("PREMVOS_BENCH_SCALING", "strong"), choices=["strong", "weak"],
                    help="strong (default): ONE synthetic video of --frames frames is shared out over the ranks exactly as the product "
                         "shards a video (premvos_amd.parallel.plan_shards: chunk-aligned contiguous ranges, boundary frame read as the "
                         "second image of a rank's last pair, ragged last chunk); a step = one pass over the video.  weak: every rank "
                         "owns --batch frames per step (rounds 1-2)")
    ap.add_argument("--frames", type=int, default=int(os.environ.get("PREMVOS_BENCH_FRAMES", "0")),
                    help="strong scaling: frame pairs of the video (default 8 chunks = 8 x --batch: one chunk per GPU of an 8-GPU node per pass)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--supplementary", default=os.environ.get("PREMVOS_BENCH_SUPPLEMENTARY", "mixed-bf16x3,1080p,1080p-mixed-bf16x3"),
                    help="after the fp32 line's timed region (N = 1, 480p, fp32 only): short passes of these further modes, measured by "
                         "this same run into the `supplementary` object -- comma-separated from {mixed-bf16x3, 1080p, 1080p-mixed-bf16x3}; 'none' skips")
    ap.add_argument("--cpu-baseline-worker", default=None, metavar="K:CORES", help=argparse.SUPPRESS)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--file-to-file", type=int, default=int(os.environ.get("PREMVOS_BENCH_F2F_FRAMES", "128")), metavar="FRAMES",
                    help="after the timed region (N = 1, 480p): also measure JPEG-in -> files-out through premvos_amd.stream on this "
                         "many synthetic frames (0 = skip; the result is the `file_to_file` object of the JSON line)")
    return ap.parse_args()


def synth_frames(batch: int, rank: int):
    """Smooth seeded noise frames + sub-pixel translated successors, uint8 RGB [B,H,W,3]."""
    from premvos_amd import synth
    return synth.video_frames(batch, H, W, rank)


def synth_boxes(batch: int, rank: int) -> torch.Tensor:
    """[B,P,4] (y0,x0,y1,x1): seeded uniform boxes with w,h in [40,400] clipped to the frame."""
    from premvos_amd import synth
    return synth.boxes(batch, P_BOXES, H, W, rank)


def _cpu_sample(cores: int):
    """One worker's sample: the oracle (plain-PyTorch restatement of the reference: kind='port') on ``cores`` threads, per stage
    3 warm-up + 10 timed runs, median.  Returns (t_flow, t_prop, t_box) in seconds."""
    from oracle import proposal_oracle as PO
    from oracle import pwc_oracle as O
    from oracle import refinement_oracle as RO
    torch.set_num_threads(cores)
    WARM, TIMED = 3, 10

    def median_time(fn):
        for _ in range(WARM):
            fn()
        ts = []
        for _ in range(TIMED):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return sorted(ts)[len(ts) // 2]

    fa, _ = synth_frames(1, 0)
    with torch.no_grad():
        sd = O.synth_state_dict(0)
        x = O.synth_frame_pair(512, 896)
        t_flow = median_time(lambda: O.pwc_forward(sd, x))
        w = PO.synth_weights(0)
        img = np.ascontiguousarray(fa[0].numpy()[:, :, ::-1])
        t_prop = median_time(lambda: PO.detect_one_image(w, img))
        rw = RO.synth_weights(0)
        boxes = synth_boxes(1, 0)[0].numpy()
        frame = fa[0].numpy()
        state = {"i": 0}

        def one_box():
            b = boxes[state["i"] % len(boxes)]
            state["i"] += 1
            net_in, crop = RO.make_input(frame, b)
            RO.output_layer(RO.deeplab_logits(rw, net_in), crop, H, W)
        t_box = median_time(one_box)
    return t_flow, t_prop, t_box


def cpu_baseline_worker(k: int, cores: int) -> None:
    """`bench.py --cpu-baseline-worker k:cores`: worker k of the whole-host CPU baseline, pinned to its own block of host threads."""
    try:
        os.sched_setaffinity(0, set(range(k * cores, (k + 1) * cores)))
    except (AttributeError, OSError):
        pass
    print(json.dumps(_cpu_sample(cores)), flush=True)


def cpu_baseline():
    """The oracle timed on the host cores on a bounded sample of the same per-frame workload, scaled to one frame.  This leg --
    and only this leg -- imports oracle/ (the checker doubles as the CPU baseline).  torch-CPU convolutions collapse when one
    process is given every thread of a 256-thread host (131 s per PWC-Net pair), so the WHOLE host is timed as
    cpu_count // 32 concurrent 32-thread processes, each pinned to its own block of threads and each running the full protocol
    at the same time (they share memory bandwidth and caches, as a CPU deployment would); value = sum over the processes of
    1 / (their seconds per frame), cores = every thread used -- opt-in, see below."""
    import subprocess
    ncpu = os.cpu_count() or 1
    per = min(ncpu, 32)
    # Default: ONE process of 32 threads.  PREMVOS_CPU_BASELINE_WHOLE_HOST=1 times cpu_count // 32 concurrent pinned processes
    # instead -- measured on the round-4 box (256 threads): the eight processes fight over memory bandwidth and L3 (PWC-Net 0.31 s
    # alone -> 9.9 s each, proposal_net 38 s), the whole host delivers 0.064 frames/s against 0.14 for the one process, and the
    # sample takes ~11 minutes -- so the better (and bounded) of the two stays the reported baseline
    nproc = max(1, ncpu // per) if os.environ.get("PREMVOS_CPU_BASELINE_WHOLE_HOST", "0") == "1" else 1
    if nproc == 1:
        samples = [_cpu_sample(per)]
    else:
        env = dict(os.environ, OMP_NUM_THREADS=str(per), MKL_NUM_THREADS=str(per), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", f"{k}:{per}"],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for k in range(nproc)]
        samples = []
        for pr in procs:
            out, _ = pr.communicate(timeout=1500)
            if pr.returncode == 0 and out.strip():
                samples.append(json.loads(out.strip().splitlines()[-1]))
        if not samples:                       # (no worker came back: fall back to the single-process sample)
            nproc, samples = 1, [_cpu_sample(per)]
    per_frame = [tf + 2 * tp + P_BOXES * tb for tf, tp, tb in samples]
    med = sorted(zip(per_frame, samples))[len(samples) // 2][1]
    return {"value": round(sum(1.0 / t for t in per_frame), 5), "unit": "frames/s", "cores": per * len(samples), "kind": "port",
            "processes": len(samples), "threads_per_process": per,
            "sample": f"{len(samples)} concurrent process(es) x {per} threads (pinned blocks of the host's {ncpu} threads), each: median of 10 "
                      f"timed runs after 3 warm-ups per stage -- 1 PWC-Net pair @512x896 ({med[0]:.2f} s), 1 proposal_net pass "
                      f"@749x1333/100 RoIs ({med[1]:.2f} s), 1 refinement box @385x385 ({med[2]:.2f} s) in the median process; fp32 "
                      f"oracle/*.py on torch {torch.__version__} CPU; a frame = flow + 2*proposal + {P_BOXES}*box = "
                      f"{sorted(per_frame)[len(per_frame) // 2]:.1f} s per process; value = sum of the processes' frames/s"}


def file_to_file(n_frames: int, chunk: int):
    """Secondary measurement, FILE TO FILE: a synthetic 480p JPEG sequence through premvos_amd.stream (one process, this GPU):
    JPEG decode, the four stages, .flo / proposal JSON / combined JSON / refined JSON with COCO-RLE strings on disk -- the
    reference's stage interface.  One cold run (plans are built), then the best of two warm runs."""
    import importlib.util
    import shutil
    import tempfile
    from PIL import Image
    from premvos_amd import stream, synth
    spec = importlib.util.spec_from_file_location("time_merge_ingest", os.path.join(ROOT, "tools", "time_merge_ingest.py"))
    tmi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tmi)
    root = tempfile.mkdtemp(prefix="premvos_f2f_")
    try:
        seq = os.path.join(root, "data", "DAVIS", "JPEGImages", "480p", "clip")
        os.makedirs(seq)
        for s0 in range(0, n_frames, 32):
            fr = synth.clip_frames(s0, min(s0 + 32, n_frames), 480, 854).numpy()
            for i, im in enumerate(fr):
                Image.fromarray(im).save(os.path.join(seq, f"{s0 + i:05d}.jpg"), quality=95)
        wd = os.path.join(root, "weights")
        os.makedirs(wd)
        torch.save({"state_dict": synth.pwc_state_dict(0)}, os.path.join(wd, "pwc.pth.tar"))
        torch.save(synth.proposal_weights(0), os.path.join(wd, "general.pt"))
        torch.save(synth.proposal_weights(1), os.path.join(wd, "specific.pt"))
        # (the refinement net's foreground bias is raised so that the masks are object-like blobs -- with the plain synthetic weights
        #  every mask is empty and the RLE / JSON side of this leg has nothing to do: tools/time_merge_ingest.py)
        torch.save(tmi.object_like_refinement_weights(), os.path.join(wd, "refine.pt"))
        out = os.path.join(root, "output", "intermediate")
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()                   # the bench's own pipeline object, still alive
        sp = stream.StreamPipeline(os.path.join(wd, "pwc.pth.tar"), os.path.join(wd, "general.pt"), os.path.join(wd, "specific.pt"),
                                   os.path.join(wd, "refine.pt"), batch=chunk, out=out)
        times = []
        for rep in range(3):
            shutil.rmtree(os.path.join(root, "output"), ignore_errors=True)
            torch.cuda.synchronize()
            t = time.perf_counter()
            n = sp.run_sequences([seq + "/"])
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t)
        assert n == n_frames
        props = sum(len(json.load(open(os.path.join(out, "combined_proposals", "clip", f"{i:05d}.json")))) for i in range(n_frames))
        files = sum(len(fs) for _, _, fs in os.walk(out))
        res = {"streaming_driver_fps": round(n_frames / min(times[1:]), 2), "cold_run_s": round(times[0], 1),
                "warm_runs_s": [round(t, 2) for t in times[1:]], "frames": n_frames, "chunk": chunk,
                "proposals_per_frame": round(props / n_frames, 1), "files_written": files, "n_gpus": 1, "measured_by_this_run": True,
                "cold_start_overhead_s": round(times[0] - min(times[1:]), 2),          # plans, buffers, first launches: cold minus warm
                "hbm_resident_gb": round((torch.cuda.max_memory_allocated() - base) / 2 ** 30, 1),      # peak of the driver's own plans, weights and buffers
                "hbm_process_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),                # ... on top of the bench's pipeline object
                "what": "python -m premvos_amd.stream's pipeline object on a synthetic 480x854 JPEG sequence (quality 95): decode -> flow + "
                        "proposals x2 + combine + refinement -> .flo / JSON / COCO-RLE files; `value` above stays the HBM-resident metric"}
        ingest = None
        if os.environ.get("PREMVOS_BENCH_MERGE_INGEST", "1") != "0":
            # the merge rank of an 8-rank gathered job on THIS GPU (tools/time_merge_ingest.py): the same pipeline object computes rank
            # 0's share while the recorded buffers of 7 other ranks arrive with every round; >= 430 frames/s = 8 x the file-to-file rate
            try:
                job = tmi.build_job(os.path.join(root, "ingest"), 64, 8, weights=False)
                ingest = tmi.measure(sp, job["clips"], 64, 8, modes=("beside",), reference=False)
                ingest["measured_by_this_run"] = True
            except Exception as e:           # noqa: BLE001 -- a secondary leg must not take the contract line down
                ingest = {"error": f"{type(e).__name__}: {e}"[:300], "measured_by_this_run": False}
            sp.out = out
        return res, ingest
    finally:
        shutil.rmtree(root, ignore_errors=True)


PEAK_BF16_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)


class BoxSampler:
    """Shader clock / socket power / temperatures of THIS rank's GPU, sampled by a host thread (amdsmi, ~20 Hz) while the timed
    region runs: the line then says under which clock and power the number was measured (the fp32 MFMA kernels run the socket
    at its 1.4 kW cap and the clock gives way by 3 ... 5 %: profiles/r05_igemm_investigation.md).  A box that does not expose
    amdsmi yields {"error": ...}; nothing else depends on it."""

    def __init__(self, index: int, period_s: float = 0.05):
        import threading
        self.samples, self.err, self._stop, self.period = [], None, threading.Event(), period_s
        try:
            import amdsmi
            self.smi = amdsmi
            try:
                amdsmi.amdsmi_init()
            except Exception:                  # noqa: BLE001 -- (already initialised by another object of this process)
                pass
            hs = amdsmi.amdsmi_get_processor_handles()
            self.h, self.matched_by = self._pick(hs, index)
            self._read()                       # fail here, not in the thread
        except Exception as e:                 # noqa: BLE001
            self.err = f"{type(e).__name__}: {e}"[:200]
        self.th = threading.Thread(target=self._loop, daemon=True)

    def _pick(self, handles, index):
        """The amdsmi handle of HIP device ``index``.  amdsmi lists every physical GPU and ignores HIP_VISIBLE_DEVICES /
        ROCR_VISIBLE_DEVICES, so the HIP index is not its position: match by PCI address (domain:bus:device) -- and refuse to
        describe some other GPU when nothing matches.  Builds that expose no PCI address fall back to the position."""
        want = None
        try:
            pr = torch.cuda.get_device_properties(index)
            want = (int(pr.pci_domain_id), int(pr.pci_bus_id), int(pr.pci_device_id))
        except Exception:                      # noqa: BLE001 -- no GPU (CPU tests) or a build without the PCI fields
            pass
        if want is None or not hasattr(self.smi, "amdsmi_get_gpu_device_bdf"):
            return handles[index % len(handles)], "position (no PCI address available)"
        seen = []
        for h in handles:
            try:
                bdf = str(self.smi.amdsmi_get_gpu_device_bdf(h))          # "0000:c1:00.0"
                dom, bus, rest = bdf.split(":")
                got = (int(dom, 16), int(bus, 16), int(rest.split(".")[0], 16))
            except Exception:                  # noqa: BLE001
                continue
            seen.append(bdf)
            if got == want:
                return h, "pci " + bdf
        raise RuntimeError(f"no amdsmi device at PCI {want[0]:04x}:{want[1]:02x}:{want[2]:02x} (HIP device {index}); amdsmi lists {seen}")

    def _read(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        clk = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 60000]
        num = lambda v: float(v) if isinstance(v, (int, float)) else None       # noqa: E731
        return (sum(clk) / len(clk) if clk else num(m.get("current_gfxclk")), min(clk) if clk else None,
                num(m.get("current_socket_power")), num(m.get("temperature_hotspot")), num(m.get("temperature_mem")),
                num(m.get("current_uclk")), num(m.get("ppt_residency_acc")), num(m.get("accumulation_counter")))

    def _loop(self):
        while not self._stop.is_set():
            try:
                self.samples.append(self._read())
            except Exception as e:             # noqa: BLE001
                self.err = f"{type(e).__name__}: {e}"[:200]
                return
            self._stop.wait(self.period)

    def __enter__(self):
        if self.err is None:
            self.th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.th.is_alive():
            self.th.join()

    def summary(self) -> dict:
        if self.err is not None and not self.samples:
            return {"error": self.err}
        col = lambda i: [s[i] for s in self.samples if s[i] is not None]        # noqa: E731
        stat = lambda v, nd=0: {"min": round(min(v), nd), "mean": round(sum(v) / len(v), nd), "max": round(max(v), nd)} if v else None   # noqa: E731
        ppt, acc = col(6), col(7)
        return {"samples": len(self.samples), "period_s": self.period, "device_matched_by": getattr(self, "matched_by", None),
                "sclk_mhz_mean_of_xcds": stat(col(0)), "sclk_mhz_slowest_xcd": stat(col(1)), "socket_power_w": stat(col(2)),
                "temperature_hotspot_c": stat(col(3)), "temperature_mem_c": stat(col(4)), "uclk_mhz": stat(col(5)),
                # share of the region the firmware reports as power-limited (PPT residency counter / accumulation counter)
                "power_limited_share": round((ppt[-1] - ppt[0]) / (acc[-1] - acc[0]), 3) if len(ppt) > 1 and len(acc) > 1 and acc[-1] > acc[0] else None,
                "source": "amdsmi_get_gpu_metrics_info on a host thread during the timed region (nominal sclk 2400 MHz, socket cap 1400 W)"}


def mfma_sustained_random(seconds: float = 1.5):
    """Pure v_mfma_f32_32x32x2_f32 on changing pseudo-random operands, back to back for ``seconds`` (after 0.5 s of settling), with the
    clock / power it ran under: the fp32 MFMA rate THIS box sustains on real data."""
    from premvos_amd import _lib
    lib = _lib.load()
    sink = torch.zeros(4, device="cuda")
    blocks, iters, st = 1024, 20000, _lib.current_stream()
    flops = blocks * 4 * iters * 16 * 4096.0

    def burst(n):
        for _ in range(n):
            lib.premvos_mfma_f32_calibrate_random(iters, blocks, sink.data_ptr(), st)
        torch.cuda.synchronize()
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        burst(4)
    n = 0
    with BoxSampler(torch.cuda.current_device()) as smi:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            burst(4)
            n += 4
        dt = time.perf_counter() - t0
    b = smi.summary()
    return {"tflops": round(n * flops / dt / 1e12, 1), "seconds": round(dt, 2),
            "sclk_mhz_mean": (b.get("sclk_mhz_mean_of_xcds") or {}).get("mean"), "socket_power_w_mean": (b.get("socket_power_w") or {}).get("mean"),
            "power_limited_share": b.get("power_limited_share")}


def hbm_ceiling():
    """What a float4 copy sustains on THIS GPU (read + write bytes / time; 1 GiB -> 1 GiB, far beyond the 256 MB Infinity Cache)."""
    from premvos_amd import _lib
    lib = _lib.load()
    n = 1 << 28                                      # floats per buffer
    src = torch.full((n,), 1.0, dtype=torch.float32, device="cuda")
    dst = torch.empty_like(src)
    st = _lib.current_stream()
    for _ in range(2):
        _lib.check(lib.premvos_hbm_copy_calibrate(src.data_ptr(), dst.data_ptr(), n // 4, st), "hbm_copy_calibrate")
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(lib.premvos_hbm_copy_calibrate(src.data_ptr(), dst.data_ptr(), n // 4, st), "hbm_copy_calibrate")
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    del src, dst
    return round(2.0 * 4 * n / (sorted(ts)[len(ts) // 2] * 1e-3) / 1e9, 1)


def roofline(pipe, batch, net_prec="fp32", flow_prec="fp32", calibrate=True):
    """Dominant kernel = conv_igemm_f32_kernel (every dense conv of the three nets).  HIP events around every one
    of its launches of one step, on the stream they are launched on; achieved = algorithmic FLOPs / time."""
    reps = 5
    items = pipe.conv_steps()          # (stage, name, fn, flops_per_step)
    samples = [[] for _ in items]
    with BoxSampler(torch.cuda.current_device()) as leg_smi:      # the clock / power THIS leg ran under (serial replay: not the timed region's)
        for _ in range(reps):
            evs = []
            for _, _, fn, _, _, _ in items:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record()
                evs.append((a, b))
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(evs):
                samples[i].append(a.elapsed_time(b))
    leg_box = leg_smi.summary()
    leg_clk = (leg_box.get("sclk_mhz_mean_of_xcds") or {}).get("mean")
    tot = [sorted(x)[len(x) // 2] for x in samples]          # median per launch: robust to a throttling transient
    # refinement launches run once per group of frames: weight their time by the calls per step
    mult = [pipe.refine_calls_per_step if st == "refine" else 1 for st, _, _, _, _, _ in items]
    ms = sum(t * m for t, m in zip(tot, mult))
    flops = sum(it[3] for it in items)
    nl = sum(mult)
    ach = flops / (ms * 1e-3) / 1e12
    per_stage = {}
    for (st, _, _, f, _, _), t, m in zip(items, tot, mult):
        a = per_stage.setdefault(st, [0.0, 0.0])
        a[0] += f
        a[1] += t * m
    # HBM bytes per launch: PMC counters cannot be read from inside this process, so the figure is the one the round's
    # rocprofv3 --pmc passes over this very command produced (tools/profile_round.sh -> profiles/rNN_conv_hbm_traffic.json,
    # method + gfx950 corrections inside); `traffic_source` says so.  null when no such file exists for this batch size.
    traffic, traffic_source = None, None
    import glob
    tfs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_conv_hbm_traffic.json")))      # newest round last
    if batch == 16 and tfs:        # the PMC passes of tools/profile_round.sh run the default batch
        try:
            traffic = round(json.load(open(tfs[-1]))["hbm_bytes_per_launch"])
            traffic_source = "profiles/" + os.path.basename(tfs[-1]) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a previous run of this command, not this run)"
        except Exception:
            traffic = None
    alg_bytes = sum(it[4] for it in items)
    # what the fp32 MFMA pipe sustains on THIS GPU with nothing else going on (pure v_mfma_f32_32x32x2_f32 loop)
    from premvos_amd import _lib
    ceiling = hbm_gbs = float("nan")
    sustained = None
    if calibrate:
        sink = torch.zeros(4, device="cuda")
        lib, blocks, iters = _lib.load(), 1024, 20000
        lib.premvos_mfma_f32_calibrate(1000, blocks, sink.data_ptr(), _lib.current_stream())
        ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ca.record()
        lib.premvos_mfma_f32_calibrate(iters, blocks, sink.data_ptr(), _lib.current_stream())
        cb.record()
        cb.synchronize()
        ceiling = blocks * 4 * iters * 16 * 4096.0 / (ca.elapsed_time(cb) * 1e-3) / 1e12
        hbm_gbs = hbm_ceiling()
        sustained = mfma_sustained_random()
    if net_prec != "fp32":
        # the optional bf16-MFMA modes (never the headline): priced against the dense bf16 peak.  bf16x3 issues three MFMAs per
        # product (hi.hi + hi.lo + lo.hi), so its issue fraction is 3x its algorithmic fraction.
        fl = {k: v[0] for k, v in per_stage.items()}
        tm = {k: v[1] for k, v in per_stage.items()}
        st = [k for k in per_stage if k != "flow" or flow_prec != "fp32"]
        a16 = sum(fl[k] for k in st) / (sum(tm[k] for k in st) * 1e-3) / 1e12
        mult = 3.0 if net_prec == "bf16x3" else 1.0
        kern = ("conv_bf16x3_s8_kernel (implicit GEMM on activations RESIDENT in the split layout S8 = {hi8, lo8} bf16 per group of 8 channels, "
                "staged by LDS-DMA, v_mfma_f32_32x32x16_bf16, three MFMAs per product, fp32 accumulate: the ResNet bottleneck chains of "
                "groups 1-3 incl. their 3x3 layers, every K >= 256 pointwise conv of the refinement net); the HBM-bound short-K layers, "
                "the RPN 3x3 (F(4x4,3x3)) and the stems stay on the fp32 kernels") if net_prec == "bf16x3" and \
            os.environ.get("PREMVOS_BF16X3_SPLIT", "1") != "0" else \
            f"conv_igemm_bf16_kernel<NPASS={int(mult)}> (fp32 activations split into bf16 hi(/lo) while staged, v_mfma_f32_32x32x16_bf16, fp32 accumulate)"
        return {"bound": "mfma", "kernel": kern + f" for the stages {st}; flow on the fp32 pipe: {flow_prec == 'fp32'}",
                "achieved": round(a16, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(a16 / PEAK_BF16_TFLOPS, 4),
                "mfma_issue_frac": round(mult * a16 / PEAK_BF16_TFLOPS, 4), "traffic": None,
                "per_stage_tflops": {k: round(v[0] / (v[1] * 1e-3) / 1e12, 1) for k, v in per_stage.items()},
                "conv_ms_per_step": round(ms, 3), "launches_per_step": nl,
                "note": "supplementary mode; the metric's line is the fp32 run (dtype f32)"}
    return {"bound": "mfma", "kernel": "the dense-conv step of the three nets on the fp32 MFMA pipe: conv_igemm_f32_kernel (implicit GEMM; "
                                       "incl. its k-slab / tail-split launches + reduce) or, where the plan-time autotuner measured it faster, "
                                       "wino_gemm_kernel + wino_output_kernel or the slab-free wino_fused_kernel (Winograd F(2x2,3x3) for 3x3 "
                                       "stride-1 layers: 2.25x fewer multiplies) or wino4_input / wino4_gemm / wino4_output (F(4x4,3x3) for K-rich "
                                       "layers: 4x fewer multiplies), so `achieved` -- ALGORITHMIC 2*M*K*N FLOPs / time -- can exceed what the pipe issues); the "
                                       "seven 1-2 channel heads run on conv_smalln_kernel and are counted with their algorithmic FLOPs",
            "winograd_layers": sum(1 for it in items if it[5].tile_hint in (2, 3, 4)),
            "winograd_f4x4_layers": sum(1 for it in items if it[5].tile_hint == 4),
            "winograd_slab_free_layers": sum(1 for it in items if it[5].tile_hint == 3),
            # FLOPs the matrix pipe actually issues (an F(2x2,3x3) layer issues 16/36 of its algorithmic 3x3 FLOPs, F(4x4,3x3) 36/144) / time / peak
            "mfma_issue_frac": round(sum(it[3] * (16.0 / 36.0 if it[5].tile_hint in (2, 3) else 0.25 if it[5].tile_hint == 4 else 1.0) for it in items) / (ms * 1e-3) / 1e12
                                     / PEAK_F32_TFLOPS, 4),
            "achieved": round(ach, 2), "peak": PEAK_F32_TFLOPS,
            "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_TFLOPS, 4),
            "frac_is": "ALGORITHMIC FLOPs / time / peak -- not pipe utilisation (Winograd layers issue fewer multiplies): read "
                       "igemm_family_frac, winograd_issued_frac and mfma_issue_frac for that",
            "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": round(alg_bytes / nl), "mfma_ceiling_measured": round(ceiling, 1),
            # ... and the same MFMA loop on CHANGING random operands, sustained for 1.5 s: the pipe's power follows the switching activity
            # of its operands; on real data the socket meets its 1.4 kW cap and the clock gives way (the constant-operand loop never does)
            "mfma_ceiling_sustained_random_data": sustained,
            # the box factor of the memory system (GB/s of a float4 copy, read + write) and ONE canonical layer of the pipeline --
            # Xception middle flow 728 -> 728 pointwise + residual at M = 100 000 pixels (xception.py:508-550), as this run timed it
            "hbm_ceiling_measured": hbm_gbs, **_canonical_layer(items, tot),
            # the split of `frac`: a Winograd layer issues 16/36 or 36/144 of its algorithmic FLOPs, so the
            # headline fraction is not pipe utilisation.  `igemm_family_frac`: the layers that multiply every algorithmic FLOP
            # (implicit GEMM incl. k-slab / tail-split launches, the streaming pointwise kernel, the 1-2 channel direct heads) --
            # algorithmic = issued there; `winograd_issued_frac`: FLOPs the Winograd layers actually issue / their time (transform
            # kernels included) / peak
            **_family_split(items, tot, mult),
            # the leg's own clock: the fp32 MFMA peak scales with the shader clock (157.3 TFLOP/s is 256 CUs x 256 FLOP / clk at 2.4 GHz),
            # and these kernels hold the socket at its 1.4 kW cap -- `frac` above stays priced against the NOMINAL peak
            "leg_box": {k: leg_box.get(k) for k in ("sclk_mhz_mean_of_xcds", "socket_power_w", "power_limited_share", "error") if leg_box.get(k) is not None},
            "peak_at_granted_clock": round(PEAK_F32_TFLOPS * leg_clk / 2400.0, 1) if leg_clk else None,
            "igemm_family_frac_at_granted_clock": (round(_family_split(items, tot, mult)["igemm_family_frac"] * 2400.0 / leg_clk, 4)
                                                   if leg_clk and _family_split(items, tot, mult)["igemm_family_frac"] else None),
            "launches_per_step": nl, "flops_per_launch": round(flops / nl, 1), "avg_launch_us": round(1e3 * ms / nl, 2),
            "conv_ms_per_step": round(ms, 3), "conv_ms_each_layer_once": round(sum(tot), 3),
            "per_stage_tflops": {k: round(v[0] / (v[1] * 1e-3) / 1e12, 1) for k, v in per_stage.items()}}


def _canonical_layer(items, tot):
    name = "conv:middle_flow/block1/unit_8/xception_module/separable_conv3_pointwise"
    for (st, n, _, f, _, d), t in zip(items, tot):
        if st == "refine" and n == name:
            return {"canonical_layer": {"what": f"{n[5:]}: {d.cin} -> {d.cout} 1x1 + residual, M = {d.n * d.ho * d.wo} pixels, one launch",
                                        "us": round(1e3 * t, 1), "tflops": round(2.0 * d.n * d.ho * d.wo * d.cin * d.cout / (t * 1e-3) / 1e12, 1)}}
    return {"canonical_layer": None}


RIDGE_F32 = PEAK_F32_TFLOPS * 1e12 / 6.29e12      # FLOP per HBM byte above which an fp32 conv is MFMA-bound (6.29 TB/s achievable)


def _family_split(items, tot, mult):
    """`frac` split by what bounds a layer.  Winograd layers issue 16/36 or 36/144 of their algorithmic FLOPs; of the layers that
    multiply every algorithmic FLOP (implicit GEMM incl. k-slab / tail-split launches, the streaming pointwise kernel, the 1-2
    channel direct heads) those whose ALGORITHMIC intensity -- 2 M K N over compulsory bytes, ops.algorithmic_bytes -- lies below
    the ridge (25 FLOP/B) cannot reach the MFMA roofline whatever the kernel does (a 64 -> 256 pointwise layer with a residual is
    14 FLOP/B: its ceiling is 0.57 of the fp32 MFMA peak): they are priced against HBM instead."""
    def issue(h):
        return 16.0 / 36.0 if h in (2, 3) else 0.25 if h == 4 else 1.0
    fl_i = t_i = fl_w = is_w = t_w = fl_m = t_m = by_h = t_h = fl_h = 0.0
    for (_, _, _, f, by, d), t, m in zip(items, tot, mult):
        if d.tile_hint in (2, 3, 4):
            fl_w += f
            is_w += f * issue(d.tile_hint)
            t_w += t * m
            continue
        fl_i += f
        t_i += t * m
        if f / max(by, 1.0) >= RIDGE_F32:
            fl_m += f
            t_m += t * m
        else:
            by_h += by
            fl_h += f
            t_h += t * m
    tf = lambda fl, t: round(fl / (t * 1e-3) / 1e12, 2) if t else None          # noqa: E731
    return {"igemm_family_tflops": tf(fl_i, t_i),
            "igemm_family_frac": round(fl_i / (t_i * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4) if t_i else None,
            "igemm_family_share_of_conv_time": round(t_i / (t_i + t_w), 4) if t_i + t_w else None,
            "igemm_mfma_bound_layers": {"tflops": tf(fl_m, t_m), "frac": round(fl_m / (t_m * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4) if t_m else None,
                                        "share_of_conv_time": round(t_m / (t_i + t_w), 4) if t_i + t_w else None},
            "igemm_hbm_bound_layers": {"what": f"algorithmic intensity < {RIDGE_F32:.0f} FLOP/B: priced against HBM",
                                       "algorithmic_GB_per_s": round(by_h / (t_h * 1e-3) / 1e9, 1) if t_h else None,
                                       "frac_of_6290_GB_per_s": round(by_h / (t_h * 1e-3) / 6.29e12, 4) if t_h else None,
                                       "tflops": tf(fl_h, t_h), "share_of_conv_time": round(t_h / (t_i + t_w), 4) if t_i + t_w else None},
            "winograd_algorithmic_tflops": tf(fl_w, t_w),
            "winograd_issued_frac": round(is_w / (t_w * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4) if t_w else None}


def supplementary_pass(mode: str, B: int, dev, passes: int = 3) -> dict:
    """One supplementary mode under this run's clock: its own pipeline object, one warm-up pass and
    ``passes`` timed passes over a synthetic video of 8 chunks (two distinct chunks of frames in HBM, walked four times), and the
    dense-conv roofline leg of that object.  ``mixed-bf16x3``: configs[2]'s arithmetic as this build reads it (PWC-Net fp32,
    proposal + refinement split-fp32 on the bf16 MFMA pipe, fp32 accumulate: plain bf16 misses north_star's 1e-3 / bit-exact-index
    bars, tests/test_gpu_error_budget.py); ``1080p``: configs[4]'s frame shape in fp32.  Never the metric's line."""
    from premvos_amd import synth
    from premvos_amd.pipeline import FramePipeline
    h, w = (1080, 1920) if mode.startswith("1080p") else (480, 854)
    net_prec, flow_prec = ("bf16x3", "fp32") if mode.endswith("mixed-bf16x3") else ("fp32", "fp32")
    t_build = time.perf_counter()
    pipe = FramePipeline(synth.pwc_state_dict(0), synth.proposal_weights(0), synth.proposal_weights(1), synth.refinement_weights(0),
                         batch=B, device=str(dev), boxes_per_frame=P_BOXES, precision=net_prec, flow_precision=flow_prec)
    clip = synth.clip_frames(0, 2 * B + 1, h, w).to(dev)
    bx = synth.clip_boxes(0, 2 * B, P_BOXES, h, w).to(dev)
    chunks = [(clip[c:c + B].contiguous(), clip[c + 1:c + B + 1].contiguous(), bx[c:c + B].contiguous()) for c in (0, B)]
    n_launch = 8

    def one_pass():
        for k in range(n_launch):
            pipe.step(*chunks[k % 2])
    one_pass()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    with BoxSampler(dev.index or 0) as smi:
        t0 = time.perf_counter()
        for _ in range(passes):
            one_pass()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    rf = roofline(pipe, B, net_prec, flow_prec, calibrate=False)
    keep = ("kernel", "achieved", "peak", "unit", "frac", "mfma_issue_frac", "igemm_family_frac", "winograd_issued_frac",
            "per_stage_tflops", "conv_ms_per_step", "launches_per_step", "note")
    box = smi.summary()
    out = {"value": round(passes * n_launch * B / dt, 2), "unit": "frames/s", "frame": f"{h}x{w}",
           "dtype": "flow f32; proposal+refinement bf16x3 (split-fp32 on the bf16 MFMA pipe, f32 accumulate)" if net_prec == "bf16x3" else "f32",
           "ms_per_launch": round(1e3 * dt / (passes * n_launch), 2), "frames_per_launch": B, "launches_timed": passes * n_launch,
           "build_and_first_pass_s": round(t_build, 1), "measured_by_this_run": True,
           "roofline": {k: rf[k] for k in keep if k in rf},
           "box": {k: box.get(k) for k in ("sclk_mhz_mean_of_xcds", "socket_power_w", "power_limited_share", "error") if box.get(k) is not None}}
    del pipe, chunks, clip, bx
    torch.cuda.empty_cache()
    return out


def self_launch(a) -> int:
    """`python bench.py --gpus N` (N > 1) outside torch.distributed: start the N ranks ourselves, one process per GPU."""
    import socket
    import subprocess
    backend = os.environ.get("PREMVOS_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < a.gpus and backend == "nccl":
        raise SystemExit(f"bench.py --gpus {a.gpus}: this node exposes {ndev} GPU(s); refusing to measure fewer GPUs than asked "
                         f"(RCCL needs one device per rank)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def shared_tune_cache(rank: int, world: int):
    """All ranks must freeze the SAME conv configurations (a different tile / k-split changes the fp32 summation order, so
    ranks would disagree in the last bits): rank 0 tunes while building and warming its pipeline, the others wait and load its
    choices.  Returns a callable rank 0 invokes once its plans exist."""
    import tempfile
    from premvos_amd import ops
    path = os.environ.get("PREMVOS_TUNE_CACHE")
    if not path:
        path = os.path.join(tempfile.gettempdir(), f"premvos_tune_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}.json")
        os.environ["PREMVOS_TUNE_CACHE"] = path
    if rank != 0:
        dist.barrier()               # rank 0 has written the cache
        ops.load_tune_cache(path)
        os.environ["PREMVOS_AUTOTUNE_FROZEN"] = "1"      # a signature rank 0 did not see is an error, not a private re-tune
        return lambda: None

    def publish():
        ops.save_tune_cache(path)
        dist.barrier()
    return publish


def main():
    a = parse()
    global H, W
    if a.cpu_baseline_worker:
        k, c = a.cpu_baseline_worker.split(":")
        return cpu_baseline_worker(int(k), int(c))
    if a.frame == "1080p":
        H, W = 1080, 1920
        a.no_cpu_baseline = True            # the CPU sample is defined on the metric's 480p workload
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} was started with WORLD_SIZE={world}")
    force_dist = os.environ.get("PREMVOS_BENCH_FORCE_DIST") == "1"     # exercise the gather path on one GPU
    use_dist = world > 1 or force_dist
    backend = os.environ.get("PREMVOS_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit(f"{world} ranks but {ndev} GPU(s) visible: one device per rank is required")
    dev = torch.device("cuda", local % max(ndev, 1))
    torch.cuda.set_device(dev)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl") is the product path; PREMVOS_BENCH_BACKEND=gloo lets the multi-rank logic be exercised by several
        # processes sharing ONE GPU (RCCL refuses two ranks on the same device)
        import datetime
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))

    from premvos_amd import synth
    from premvos_amd.parallel import ResultExchange
    from premvos_amd.pipeline import FramePipeline

    B = a.batch
    prec = a.precision
    flow_prec = "fp32" if prec.startswith("mixed") else prec
    net_prec = prec.split("-")[1] if prec.startswith("mixed") else prec
    publish = shared_tune_cache(rank, world) if world > 1 else (lambda: None)
    # cold start of this rank = building the pipeline object from weights that exist (packing, uploads) + its first launch (plans,
    # arenas, table lookup, graph capture).  Round 5's 18 ... 19 s counted 6.8 s of drawing the SYNTHETIC weights and ~6 s of
    # synthesising the 257 frames of the video on the host into it (tools/dev/cold_start_profile.py): both are inputs a product run
    # reads from disk; they are timed apart (`synthetic_inputs_s`)
    t_syn = time.perf_counter()
    weights = (synth.pwc_state_dict(0), synth.proposal_weights(0), synth.proposal_weights(1), synth.refinement_weights(0))
    synthetic_inputs_s = time.perf_counter() - t_syn
    t_proc = time.perf_counter()
    pipe = FramePipeline(*weights, batch=B, device=str(dev), boxes_per_frame=P_BOXES, precision=net_prec, flow_precision=flow_prec)
    del weights
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t_proc
    t_syn = time.perf_counter()
    strong = a.scaling == "strong"
    # strong scaling: the documented default video is 16 chunks (256 frame pairs at --batch 16): TWO launches per rank and pass at
    # N = 8, so that one slow launch is not the whole step; --frames overrides
    T = (a.frames if a.frames > 0 else 16 * B) if strong else B * world
    xchg = ResultExchange(B, H, W, P_BOXES, dev) if use_dist else None
    if strong:
        # the product's own sharding of ONE video (premvos_amd.stream.run): chunk-aligned frame ranges; frame pair t = (t, t+1)
        # belongs to the owner of frame t, so a rank also reads frame `end` (the video has T + 1 frames: T pairs)
        from premvos_amd.parallel import plan_shards
        plans = [plan_shards([T], world, r, B) for r in range(world)]
        mine = plans[rank]
        chunks = []                                     # (frames_a, frames_b, boxes, valid frames) resident in HBM
        for _, s0, e0 in mine:
            clip = synth.clip_frames(s0, e0 + 1, H, W).to(dev)
            cb = synth.clip_boxes(s0, e0, P_BOXES, H, W).to(dev)
            for c0 in range(0, e0 - s0, B):
                n = min(B, e0 - s0 - c0)
                ca, cbx, cbb = clip[c0:c0 + n], cb[c0:c0 + n], clip[c0 + 1:c0 + n + 1]
                if n < B:                               # ragged last chunk of the video: padded to the launch batch (its work is
                    pad = B - n                         # timed, its frames are not counted)
                    ca = torch.cat([ca, ca[-1:].expand(pad, -1, -1, -1)]).contiguous()
                    cbb = torch.cat([cbb, cbb[-1:].expand(pad, -1, -1, -1)]).contiguous()
                    cbx = torch.cat([cbx, cbx[-1:].expand(pad, -1, -1)]).contiguous()
                chunks.append((ca.contiguous(), cbb.contiguous(), cbx.contiguous(), n))
        rounds = max(sum(-(-(e0 - s0) // B) for _, s0, e0 in p) for p in plans)     # exchanges per pass (ranks stay in step)
        if not chunks:                                  # more ranks than chunks: this rank only takes part in the gathers
            z = synth.clip_frames(0, 2, H, W).to(dev)
            chunks_dummy = (z[:1].expand(B, -1, -1, -1).contiguous(), z[1:].expand(B, -1, -1, -1).contiguous(),
                            synth.clip_boxes(0, 1, P_BOXES, H, W).expand(B, -1, -1).contiguous().to(dev), 0)
        fa, fb, boxes = (chunks[0] if chunks else chunks_dummy)[:3]
        last = {"r": None}

        def step():                                     # one pass over the video: this rank's chunks + one gather per round
            for k in range(rounds):
                if k < len(chunks):
                    last["r"] = pipe.step(*chunks[k][:3])
                if use_dist:                            # async: the gather of round k overlaps the compute of round k + 1
                    xchg.exchange_async(last["r"])
            return last["r"]
    else:
        fa, fb = synth_frames(B, rank)
        fa, fb = fa.to(dev), fb.to(dev)
        boxes = synth_boxes(B, rank).to(dev)

        def step():
            r = pipe.step(fa, fb, boxes)
            if use_dist:        # the single exchange of the path: ONE gather of one packed buffer -> merge rank
                xchg.exchange_async(r)
            return r

    torch.cuda.synchronize()
    synthetic_inputs_s += time.perf_counter() - t_syn
    t_launch = time.perf_counter()
    t_first = None
    if world > 1 and rank == 0:
        pipe.step(fa, fb, boxes)          # builds + tunes every plan (no collective: the other ranks wait in publish())
        torch.cuda.synchronize()
        t_first = time.perf_counter()
    t_w = time.perf_counter()
    publish()
    waited = time.perf_counter() - t_w if world > 1 and rank != 0 else 0.0      # (for rank 0's tuning: not this rank's cold start)
    if t_first is None:
        r0 = pipe.step(fa, fb, boxes)     # first launch of this rank: plans, graphs (and something to pack for the gathers a
        if strong:                        # chunk-less rank only takes part in)
            last["r"] = r0
        torch.cuda.synchronize()
        t_first = time.perf_counter()
    cold_start_s = build_s + (t_first - t_launch - waited)     # pipeline object from existing weights + the first launch
    smi = BoxSampler(dev.index or 0)
    for _ in range(a.warmup):
        step()
    if use_dist:
        xchg.flush()
        dist.barrier()
        xchg.wait_s = 0.0
    torch.cuda.synchronize()
    with smi:
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        if use_dist:
            xchg.flush()                      # every gather has landed on the merge rank inside the timed region
        torch.cuda.synchronize()
        dt_own = time.perf_counter() - t0     # this rank's own work + its exchanges, before it waits for the others
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    per_rank = None
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if backend == "gloo" else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # scaling diagnostics (no curve is claimed from them): what every rank spent on its own, how long it sat in
        # ResultExchange.wait / flush, how far apart the ranks finished, and its cold start
        mine = {"rank": rank, "ms_per_step": round(1e3 * dt_own / a.steps, 3), "launches_per_step": len(chunks) if strong else 1,
                "exchange_wait_ms_per_step": round(1e3 * xchg.wait_s / a.steps, 3), "cold_start_s": round(cold_start_s, 2),
                "sclk_mhz_mean": ((smi.summary().get("sclk_mhz_mean_of_xcds") or {}).get("mean") if smi.err is None else None)}
        rows = [None] * world
        dist.all_gather_object(rows, mine)
        own = [r["ms_per_step"] for r in rows]
        per_rank = {"ms_per_step": own, "launches_per_step": [r["launches_per_step"] for r in rows],
                    "exchange_wait_ms_per_step": [r["exchange_wait_ms_per_step"] for r in rows],
                    "cold_start_s": [r["cold_start_s"] for r in rows], "sclk_mhz_mean": [r["sclk_mhz_mean"] for r in rows],
                    "rank_skew_ms_per_step": round(max(own) - min(own), 3),
                    "slowest_rank": int(max(range(world), key=lambda k: own[k])),
                    "what": "ms_per_step: a rank's own launches + its gathers (before the closing barrier); exchange_wait: time blocked "
                            "in ResultExchange.wait / flush (a gather that did not finish under the next chunk's compute); skew = "
                            "slowest - fastest rank; `ms_per_step` of the line is the max over ranks incl. the barrier"}

    frames = a.steps * T
    out = {
        "metric": "frames/sec (proposal+refine+flow) on 480p DAVIS frames",
        "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True,
        "scaling": a.scaling, "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3 (split-fp32 on the bf16 MFMA pipe, f32 accumulate)",
                  "mixed-bf16x3": "flow f32; proposal+refinement bf16x3 (split-fp32, f32 accumulate)",
                  "mixed-bf16": "flow f32; proposal+refinement bf16 (f32 accumulate)"}[prec],
        "data": "synthetic",
        "box": smi.summary(),          # clock / power / temperature of rank 0's GPU over the timed region
        "config": {"workload": ("configs[3] on one node: full per-frame pipe on synthetic DAVIS-shape 480x854 uint8 frames in "
                                "HBM: PWC-Net flow (512x896) + proposal_net x2 weight sets (749x1333, ResNet-101-C4, 100 RoIs) "
                                if a.frame == "480p" else
                                "configs[4] shape on one node (supplementary, fp32): synthetic 1080x1920 uint8 frames in HBM: "
                                "PWC-Net flow (1088x1920) + proposal_net x2 weight sets (750x1333, ResNet-101-C4, 100 RoIs) ")
                               + f"+ refinement_net (Xception-65 DeepLabv3+, {P_BOXES} seeded boxes/frame @385x385); conv arithmetic: {prec}; "
                               + "results (flow, masks, conf, boxes) left in HBM",
                   "frames_per_step": T, "frames_per_launch_per_gpu": B,
                   "step": (f"one pass over ONE synthetic video of {T} frame pairs = {-(-T // B)} chunks of {B} frames, chunk ranges "
                            f"shared out over the ranks by premvos_amd.parallel.plan_shards (rank r also reads the first frame of "
                            f"rank r+1's range; a ragged last chunk is padded to {B} and only its real frames are counted)"
                            if strong else f"every rank runs its own {B} frames"),
                   "stages": ["flow", "proposal_general", "proposal_specific", "refinement"],
                   "gflop_per_frame": 2420 if a.frame == "480p" else 3018,
                   "parallelism": f"frames sharded over {world} GPU(s), no data-path collective; ONE asynchronous "
                                  f"{'RCCL' if backend == 'nccl' else backend} gather per chunk of one packed buffer per rank "
                                  f"({xchg.nbytes if xchg else 0} B: flow f32, masks bit-packed, boxes/scores/conf) to rank 0"},
    }
    if per_rank is not None:
        out["per_rank"] = per_rank
    out["cold_start_s"] = round(cold_start_s, 2)
    out["cold_start"] = {"build_pipeline_s": round(build_s, 2), "first_launch_s": round(cold_start_s - build_s, 2),
                         "synthetic_inputs_s": round(synthetic_inputs_s, 2),
                         "what": "cold_start_s = pipeline object from weights in host memory (packing on the device, uploads) + first launch "
                                 "(plans, arenas, shipped-table lookup, graph capture); synthetic_inputs_s = drawing the random weights and "
                                 "the synthetic video on the host (not part of a product run; counted into cold_start_s up to round 5)"}
    if rank == 0:
        from premvos_amd import ops
        out["conv_configurations"] = ops.tune_info()     # which table / rule froze the kernels (reproducibility)
        if not a.no_roofline:
            out["roofline"] = rf = roofline(pipe, B, net_prec, flow_prec)
            # ... beside the per-layer replay above (each layer back to back with itself: warm L2 / MALL, one stream): the WHOLE launch as
            # the timed region ran it -- every kernel of B frames (depthwise, RoIAlign, cost volume, resizes, NMS ... included), five
            # streams -- priced against the same peak: B x gflop_per_frame / (this rank's seconds per launch)
            n_launch = len(chunks) if strong else 1
            if n_launch and net_prec == "fp32":
                ms_launch = 1e3 * dt_own / a.steps / n_launch
                gf = out["config"]["gflop_per_frame"]
                peak = rf["peak"]
                rf["whole_step_ms_per_launch"] = round(ms_launch, 3)
                rf["whole_step_tflops"] = round(B * gf * 1e9 / (ms_launch * 1e-3) / 1e12, 2)
                rf["whole_step_frac"] = round(B * gf * 1e9 / (ms_launch * 1e-3) / 1e12 / peak, 4)
                rf["non_conv_ms_per_launch"] = round(ms_launch - rf["conv_ms_per_step"], 3)
                rf["whole_step_is"] = ("B x config.gflop_per_frame (algorithmic, all kernels) / this rank's wall time per launch in the timed "
                                       "region / peak; non_conv_ms_per_launch = that wall time - conv_ms_per_step (the serial replay of the "
                                       "dense convs): depthwise, RoIAlign, cost volume, resizes, selection kernels and what the five-stream "
                                       "overlap does not hide")
        # secondary: the same path measured FILE TO FILE (JPEG decode, .flo / JSON / COCO-RLE writing included) by the streaming
        # driver -- by THIS run at N = 1, after the timed region; `value` above is the HBM-resident metric, never this
        if world == 1 and a.frame == "480p" and a.file_to_file > 0 and prec == "fp32":
            try:
                out["file_to_file"], ingest = file_to_file(a.file_to_file, int(os.environ.get("PREMVOS_STREAM_BATCH", "8")))
                if ingest is not None:
                    out["merge_ingest"] = ingest
            except Exception as e:           # noqa: BLE001 -- a secondary leg must not take the contract line down
                out["file_to_file"] = {"error": f"{type(e).__name__}: {e}"[:300], "measured_by_this_run": False}
        # supplementary modes under the same run's clock (never `value`): the fp32 object is freed first
        modes = [m for m in a.supplementary.split(",") if m in ("mixed-bf16x3", "1080p", "1080p-mixed-bf16x3")]
        if world == 1 and a.frame == "480p" and prec == "fp32" and modes:
            pipe = step = fa = fb = boxes = None
            if strong:
                chunks.clear()
                last["r"] = None
            torch.cuda.empty_cache()
            out["supplementary"] = {}
            for m in modes:
                try:
                    out["supplementary"][m] = supplementary_pass(m, B, dev)
                except Exception as e:       # noqa: BLE001 -- a secondary leg must not take the contract line down
                    out["supplementary"][m] = {"error": f"{type(e).__name__}: {e}"[:300], "measured_by_this_run": False}
                    torch.cuda.empty_cache()
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
