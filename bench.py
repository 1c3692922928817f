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
Each rank owns its own frames (no data-path collective); per step the results (flow, masks, conf, boxes) are
gathered to rank 0 -- the merge rank -- with RCCL, inside the timed region.  Rank 0 prints ONE JSON line.
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def synth_frames(batch: int, rank: int):
    """Smooth seeded noise frames + sub-pixel translated successors, uint8 RGB [B,H,W,3]."""
    from oracle import pwc_oracle as O
    a, b = [], []
    for i in range(batch):
        pair = O.synth_frame_pair(H, W + 2, seed=1234 + 100 * rank + i, shift=(1.5 + 0.25 * i, -0.75))
        fr = (pair[0, :, :, :W].permute(1, 2, 0) * 255).round().to(torch.uint8)       # [H,W,6]
        a.append(fr[..., :3])
        b.append(fr[..., 3:])
    return torch.stack(a).contiguous(), torch.stack(b).contiguous()


def synth_boxes(batch: int, rank: int) -> torch.Tensor:
    """[B,P,4] (y0,x0,y1,x1): seeded uniform boxes with w,h in [40,400] clipped to the frame."""
    rng = np.random.default_rng(4321 + rank)
    wh = rng.uniform(40, 400, (batch, P_BOXES, 2))
    wh = np.minimum(wh, [W, H])
    xy = rng.uniform(0, 1, (batch, P_BOXES, 2)) * (np.array([W, H]) - wh)
    return torch.tensor(np.stack([xy[..., 1], xy[..., 0], xy[..., 1] + wh[..., 1], xy[..., 0] + wh[..., 0]], -1),
                        dtype=torch.float32)


def cpu_baseline():
    """The oracle (plain-PyTorch restatement of the reference: kind='port') timed on the host cores on a bounded
    sample of the same per-frame workload, scaled to one frame."""
    from oracle import proposal_oracle as PO
    from oracle import pwc_oracle as O
    from oracle import refinement_oracle as RO
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 32)          # torch-CPU convs collapse when oversubscribed (256 threads: 131 s per PWC pair)
    torch.set_num_threads(cores)
    fa, fb = synth_frames(1, 0)
    with torch.no_grad():
        sd = O.synth_state_dict(0)
        x = O.synth_frame_pair(512, 896)
        O.pwc_forward(sd, x)
        t = time.perf_counter(); O.pwc_forward(sd, x); t_flow = time.perf_counter() - t
        w = PO.synth_weights(0)
        img = np.ascontiguousarray(fa[0].numpy()[:, :, ::-1])
        t = time.perf_counter(); PO.detect_one_image(w, img); t_prop = time.perf_counter() - t
        rw = RO.synth_weights(0)
        boxes = synth_boxes(1, 0)[0].numpy()
        nb = 3
        t = time.perf_counter()
        for i in range(nb):
            net_in, crop = RO.make_input(fa[0].numpy(), boxes[i])
            RO.output_layer(RO.deeplab_logits(rw, net_in), crop, H, W)
        t_box = (time.perf_counter() - t) / nb
    per_frame = t_flow + 2 * t_prop + P_BOXES * t_box
    return {"value": round(1.0 / per_frame, 5), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 PWC-Net pair @512x896 ({t_flow:.2f} s) + 1 proposal_net pass @749x1333/100 RoIs ({t_prop:.2f} s) "
                      f"+ {nb} refinement boxes @385x385 ({t_box:.2f} s/box), fp32 oracle/*.py on torch "
                      f"{torch.__version__} CPU with {cores} of {ncpu} host threads; scaled to a frame as "
                      f"flow + 2*proposal + {P_BOXES}*box = {per_frame:.1f} s"}


def roofline(pipe, batch):
    """Dominant kernel = conv_igemm_f32_kernel (every dense conv of the three nets).  HIP events around every one
    of its launches of one step, on the stream they are launched on; achieved = algorithmic FLOPs / time."""
    reps = 5
    items = pipe.conv_steps()          # (stage, name, fn, flops_per_step)
    samples = [[] for _ in items]
    for _ in range(reps):
        evs = []
        for _, _, fn, _, _ in items:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(evs):
            samples[i].append(a.elapsed_time(b))
    tot = [sorted(x)[len(x) // 2] for x in samples]          # median per launch: robust to a throttling transient
    # refinement launches run once per group of frames: weight their time by the calls per step
    mult = [pipe.refine_calls_per_step if st == "refine" else 1 for st, _, _, _, _ in items]
    ms = sum(t * m for t, m in zip(tot, mult))
    flops = sum(it[3] for it in items)
    nl = sum(mult)
    ach = flops / (ms * 1e-3) / 1e12
    per_stage = {}
    for (st, _, _, f, _), t, m in zip(items, tot, mult):
        a = per_stage.setdefault(st, [0.0, 0.0])
        a[0] += f
        a[1] += t * m
    traffic = None      # HBM bytes per launch from the PMC passes (profiles/r01_conv_hbm_traffic.json: method + corrections)
    import glob
    tfs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_conv_hbm_traffic.json")))      # newest round last
    if batch == 16 and tfs:        # the PMC passes of tools/profile_round.sh run the default batch
        try:
            traffic = round(json.load(open(tfs[-1]))["hbm_bytes_per_launch"])
        except Exception:
            traffic = None
    alg_bytes = sum(it[4] for it in items)
    # what the fp32 MFMA pipe sustains on THIS GPU with nothing else going on (pure v_mfma_f32_32x32x2_f32 loop)
    from premvos_amd import _lib
    sink = torch.zeros(4, device="cuda")
    lib, blocks, iters = _lib.load(), 1024, 20000
    lib.premvos_mfma_f32_calibrate(1000, blocks, sink.data_ptr(), _lib.current_stream())
    ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ca.record()
    lib.premvos_mfma_f32_calibrate(iters, blocks, sink.data_ptr(), _lib.current_stream())
    cb.record()
    cb.synchronize()
    ceiling = blocks * 4 * iters * 16 * 4096.0 / (ca.elapsed_time(cb) * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "conv_igemm_f32_kernel (every dense conv of the three nets; the seven 1-2 channel heads run "
                                       "on conv_smalln_kernel and are counted with their algorithmic FLOPs)", "achieved": round(ach, 2), "peak": PEAK_F32_TFLOPS,
            "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_TFLOPS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": round(alg_bytes / nl), "mfma_ceiling_measured": round(ceiling, 1),
            "launches_per_step": nl, "flops_per_launch": round(flops / nl, 1), "avg_launch_us": round(1e3 * ms / nl, 2),
            "conv_ms_per_step": round(ms, 3), "conv_ms_each_layer_once": round(sum(tot), 3),
            "per_stage_tflops": {k: round(v[0] / (v[1] * 1e-3) / 1e12, 1) for k, v in per_stage.items()}}


def main():
    a = parse()
    global H, W
    if a.frame == "1080p":
        H, W = 1080, 1920
        a.no_cpu_baseline = True            # the CPU sample is defined on the metric's 480p workload
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("PREMVOS_BENCH_FORCE_DIST") == "1"     # exercise the gather path on one GPU
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local if local < torch.cuda.device_count() else 0)
        # RCCL ("nccl") is the product path; PREMVOS_BENCH_BACKEND=gloo lets the multi-rank logic be exercised by several
        # processes sharing ONE GPU (RCCL refuses two ranks on the same device)
        dist.init_process_group(os.environ.get("PREMVOS_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    assert world == a.gpus or world == 1, (world, a.gpus)
    dev = torch.device("cuda", local if local < torch.cuda.device_count() else 0)
    torch.cuda.set_device(dev)

    # oracle modules are used here ONLY as generators of synthetic weights / frames (no checkpoint, no DAVIS)
    from oracle import proposal_oracle as PO
    from oracle import pwc_oracle as O
    from oracle import refinement_oracle as RO
    from premvos_amd.pipeline import FramePipeline

    B = a.batch
    prec = a.precision
    flow_prec = "fp32" if prec.startswith("mixed") else prec
    net_prec = prec.split("-")[1] if prec.startswith("mixed") else prec
    pipe = FramePipeline(O.synth_state_dict(0), PO.synth_weights(0), PO.synth_weights(1), RO.synth_weights(0),
                         batch=B, device=str(dev), boxes_per_frame=P_BOXES, precision=net_prec, flow_precision=flow_prec)
    fa, fb = synth_frames(B, rank)
    fa, fb = fa.to(dev), fb.to(dev)
    boxes = synth_boxes(B, rank).to(dev)

    gbuf = None
    use_dist = world > 1 or force_dist
    if use_dist and rank == 0:
        gbuf = {"flow": [torch.empty((B, H, W, 2), dtype=torch.float32, device=dev) for _ in range(world)],
                "masks": [torch.empty((B, P_BOXES, H, W), dtype=torch.uint8, device=dev) for _ in range(world)],
                "small": [torch.empty((B, 2 * 20 * 5 + P_BOXES + 2), dtype=torch.float32, device=dev) for _ in range(world)]}

    def step():
        r = pipe.step(fa, fb, boxes)
        if use_dist:        # the single exchange of the path: results -> merge rank
            small = torch.cat([r["general_boxes"].reshape(B, -1), r["general_probs"], r["specific_boxes"].reshape(B, -1),
                               r["specific_probs"], r["conf"], r["general_count"].view(B, 1).float(),
                               r["specific_count"].view(B, 1).float()], 1).contiguous()
            dist.gather(r["flow"], gbuf["flow"] if rank == 0 else None, dst=0)
            dist.gather(r["masks"], gbuf["masks"] if rank == 0 else None, dst=0)
            dist.gather(small, gbuf["small"] if rank == 0 else None, dst=0)
        return r

    for _ in range(a.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    frames = a.steps * B * world
    out = {
        "metric": "frames/sec (proposal+refine+flow) on 480p DAVIS frames",
        "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3 (split-fp32 on the bf16 MFMA pipe, f32 accumulate)",
                  "mixed-bf16x3": "flow f32; proposal+refinement bf16x3 (split-fp32, f32 accumulate)",
                  "mixed-bf16": "flow f32; proposal+refinement bf16 (f32 accumulate)"}[prec],
        "data": "synthetic",
        "config": {"workload": ("configs[3] on one node: full per-frame pipe on synthetic DAVIS-shape 480x854 uint8 frames in "
                                "HBM: PWC-Net flow (512x896) + proposal_net x2 weight sets (749x1333, ResNet-101-C4, 100 RoIs) "
                                if a.frame == "480p" else
                                "configs[4] shape on one node (supplementary, fp32): synthetic 1080x1920 uint8 frames in HBM: "
                                "PWC-Net flow (1088x1920) + proposal_net x2 weight sets (750x1333, ResNet-101-C4, 100 RoIs) ")
                               + f"+ refinement_net (Xception-65 DeepLabv3+, {P_BOXES} seeded boxes/frame @385x385); conv arithmetic: {prec}; "
                               + "results (flow, masks, conf, boxes) left in HBM",
                   "frames_per_step_per_gpu": B, "stages": ["flow", "proposal_general", "proposal_specific", "refinement"],
                   "gflop_per_frame": 2420 if a.frame == "480p" else 3018,
                   "parallelism": f"frames sharded over {world} GPU(s); RCCL gather of results to rank 0 per step"},
    }
    if rank == 0:
        if not a.no_roofline:
            out["roofline"] = roofline(pipe, B)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
