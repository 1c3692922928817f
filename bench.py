#!/usr/bin/env python
"""bench.py -- the driver contract.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic 480x854 frames resident in HBM.
Each rank owns its own shard of frames (no data-path collective); the per-chunk results are
gathered to rank 0 (the merge rank) with one RCCL gather, inside the timed region.  Rank 0 prints
ONE JSON line.  value = frames processed by all ranks / max-over-ranks wall time.
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

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 matrix/vector peak
H, W = 480, 854              # DAVIS 480p
H_, W_ = 512, 896            # script_pwc_multi.py:38-45 (multiples of 64)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PREMVOS_BENCH_BATCH", "4")),
                    help="frame pairs per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=6)
    return ap.parse_args()


def cpu_baseline(iters: int):
    """The oracle (plain-PyTorch restatement of the reference, kind='port') on the host cores."""
    from oracle import pwc_oracle as O
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 32)          # torch-CPU convs collapse when oversubscribed (256 threads: 131 s/pair)
    torch.set_num_threads(cores)
    sd = O.synth_state_dict(0)
    x = O.synth_frame_pair(H_, W_)
    budget, ts = 20.0, []
    with torch.no_grad():
        t_all = time.perf_counter()
        O.pwc_forward(sd, x)                      # warm-up
        if time.perf_counter() - t_all > budget:  # pathological host: report the single sample
            ts.append(time.perf_counter() - t_all)
        while len(ts) < iters and time.perf_counter() - t_all < budget:
            t = time.perf_counter()
            O.pwc_forward(sd, x)
            ts.append(time.perf_counter() - t)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{len(ts)} PWC-Net forwards at 1x6x{H_}x{W_} fp32 (oracle/pwc_oracle.py), median, "
                      f"bounded to ~{int(budget)} s; torch {torch.__version__} CPU, {cores} of {ncpu} host threads"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == a.gpus or world == 1, (world, a.gpus)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from oracle import pwc_oracle as O                      # synthetic weights/frames generator only
    from premvos_amd.flow.driver import FlowStage

    stage = FlowStage(O.synth_state_dict(0), batch=a.batch, device=str(dev))
    # synthetic uint8 frames of DAVIS shape, resident in HBM before the clock starts
    g = torch.Generator().manual_seed(1234 + rank)
    pair = O.synth_frame_pair(H, W - W % 2, seed=1234 + rank)   # smooth field + sub-pixel motion
    fr = torch.zeros((2, H, W, 3), dtype=torch.uint8)
    for f in range(2):
        img = (pair[0, 3 * f:3 * f + 3].permute(1, 2, 0) * 255).round().to(torch.uint8)
        fr[f, :, :img.shape[1]] = img
        fr[f, :, img.shape[1]:] = img[:, -1:]
    frames1 = fr[0].unsqueeze(0).repeat(a.batch, 1, 1, 1).contiguous().to(dev)
    frames2 = fr[1].unsqueeze(0).repeat(a.batch, 1, 1, 1).contiguous().to(dev)
    del g

    gather_buf = None
    if world > 1 and rank == 0:
        gather_buf = [torch.empty((a.batch, H, W, 2), dtype=torch.float32, device=dev) for _ in range(world)]

    def step():
        flo = stage.run(frames1, frames2)                  # [B,H,W,2] fp32, the .flo payloads
        if world > 1:
            dist.gather(flo, gather_buf, dst=0)
        return flo

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    frames = a.steps * a.batch * world
    out = {
        "metric": "frames/sec (proposal+refine+flow) on 480p DAVIS frames",
        "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: PWC-Net flow on synthetic 480x854 frame pairs "
                               f"(resized to {H_}x{W_}), fp32, uint8 frames in HBM -> .flo payload in HBM; "
                               "proposal/refinement stages not yet in the timed path (round 1)",
                   "pairs_per_step_per_gpu": a.batch, "stages": ["flow"],
                   "parallelism": f"frames sharded over {world} GPU(s), one RCCL gather per step to rank 0"},
    }
    if rank == 0:
        out["roofline"] = stage.roofline(frames1, frames2, PEAK_F32_TFLOPS)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_iters)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
