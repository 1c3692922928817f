#!/usr/bin/env python
"""The serial bound SURVEY 8(e) / 8(f1) predict: MergeTrack's per-video loop is sequential in t
(MergeTrack/merge.py:76-108) and, every frame, warps the tracked objects' masks by the flow (merge_functions.py:209-241),
runs refinement_net on the <= ~10 warped boxes (refinement_net_functions.py:38-64), scores them against the frame's proposals
by mask IoU (merge_functions.py:38-45) and RLE-encodes what it keeps.  MergeTrack itself stays the reference's; this tool times
a MERGE-SHAPED loop built from the package's 8(f1) helpers, results of frame t feeding frame t + 1, masks resident in HBM:

    per frame:  warp_proposals(tracked, flow_t, device_masks=True)  ->  refine the warped boxes (one batched pass, P-bucket plan)
                ->  mask_iou(refined, candidates_t)  ->  encode_masks(refined)  ->  tracked := refined

It reports frames/s of ONE video (the loop cannot be sharded over frames), the per-phase split, and what that implies for a
node: with V videos in flight the merge stage scales "replicas only" (one loop per video, at most one per GPU), so the node-level
cap is min(8, V) x this rate -- to be compared with the 8 x ~53 frames/s the batch stages produce.  The refinement pass is timed
eager (the ~400 launches of the net issued one by one) and as a replayed HIP graph per box bucket (the product's default).

    python tools/time_merge_loop.py [--frames 64] [--objects 10] [--candidates 20] [--out gpurun_out/merge_loop.json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--objects", type=int, default=10, help="tracked objects = warped boxes refined per frame (DAVIS: <= ~10)")
    ap.add_argument("--candidates", type=int, default=20, help="proposals of the frame the refined masks are scored against")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "merge_loop.json"))
    a = ap.parse_args()
    from premvos_amd import mergetrack as mt
    from premvos_amd import rle, synth
    from premvos_amd.refinement import RefinementNet
    from premvos_amd.refinement.driver import _bucket
    H, W, T, N = 480, 854, a.frames, a.objects
    dev = "cuda"
    frames = synth.clip_frames(0, T + 1, H, W).to(dev)
    rng = np.random.default_rng(5)
    # a smooth synthetic flow per frame (a few pixels, sub-pixel parts) resident in HBM, as the flow stage leaves it
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    flows = torch.from_numpy(np.stack([np.stack([2.5 * np.sin(yy / 97.0 + 0.1 * t) + 1.25, 1.5 * np.cos(xx / 131.0 - 0.07 * t) - 0.5], -1)
                                       for t in range(T)]).astype(np.float32)).to(dev)
    # candidates of every frame: seeded box masks (what decoding the frame's refined-proposal RLEs gives MergeTrack)
    cand_boxes = synth.clip_boxes(0, T, a.candidates, H, W).numpy()

    def box_masks(boxes):
        m = torch.zeros((len(boxes), H, W), dtype=torch.uint8, device=dev)
        for i, (y0, x0, y1, x1) in enumerate(boxes):
            m[i, int(y0):max(int(y1), int(y0) + 1), int(x0):max(int(x1), int(x0) + 1)] = 1
        return m
    cands = [box_masks(cand_boxes[t]) for t in range(T)]
    start = box_masks(synth.boxes(1, N, H, W, rank=99)[0].numpy())
    results = {}
    for mode in ("eager", "graph"):
        net = RefinementNet(synth.refinement_weights(0), 16, dev, use_graph=(mode == "graph"))
        tracked = [{"mask": start[i], "final_score": 0.5, "object_score": 0.5, "id": i} for i in range(N)]
        phases = {"warp+rle+bbox": 0.0, "refine": 0.0, "iou": 0.0, "encode": 0.0}
        launches = None

        def one_frame(t, tracked, timed):
            def tick(name, t0):
                if timed:
                    torch.cuda.synchronize()
                    phases[name] += time.perf_counter() - t0
                return time.perf_counter()
            t0 = time.perf_counter()
            warped = mt.warp_proposals(tracked, flows[t], device_masks=True)                 # merge_functions.py:219-241
            t0 = tick("warp+rle+bbox", t0)
            boxes = np.array([[b[1], b[0], b[1] + b[3], b[0] + b[2]] for b in (w["bbox"] for w in warped)], np.float32)
            p = net.refine(frames[t + 1], torch.from_numpy(boxes).to(dev), max_boxes=_bucket(len(boxes)))   # do_refinement
            conf = p.conf[:len(boxes)].cpu().numpy()
            refined = p.mask[:len(boxes)].clone()
            t0 = tick("refine", t0)
            iou = mt.mask_iou(refined, cands[t])                                             # merge_functions.py:38-45
            t0 = tick("iou", t0)
            segs = mt.encode_masks(refined)                                                  # what the loop writes / keeps
            tick("encode", t0)
            out = []
            for i, w in enumerate(warped):
                m = refined[i]
                if int(rle.area(segs[i])) == 0:                                              # an object that left the frame: re-seed
                    m = start[i]
                out.append({"mask": m, "final_score": float(np.clip(conf[i], -1, 1)), "object_score": float(iou[i].max()), "id": w["id"]})
            return out
        for t in range(3):                                   # warm-up: plans / graphs of the bucket, allocator
            tracked = one_frame(t, tracked, False)
        if mode == "eager":
            launches = len(net.plan(_bucket(N), H, W, False, 0).steps)
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        for t in range(T):
            tracked = one_frame(t, tracked, False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t_all
        for t in range(min(T, 32)):                          # a second, instrumented pass for the phase split (syncs added)
            tracked = one_frame(t, tracked, True)
        n_ph = min(T, 32)
        results[mode] = {"frames_per_s_one_video": round(T / dt, 2), "ms_per_frame": round(1e3 * dt / T, 3),
                         "phase_ms_per_frame_with_syncs": {k: round(1e3 * v / n_ph, 3) for k, v in phases.items()},
                         "launches_per_refinement_pass": launches}
        del net
        torch.cuda.empty_cache()
    best = max(r["frames_per_s_one_video"] for r in results.values())
    out = {"what": "merge-shaped loop, sequential in t, one synthetic 480x854 video: warp_proposals (resident masks) -> refinement of the "
                   f"{N} warped boxes -> mask_iou vs {a.candidates} candidates -> encode_masks; results feed frame t + 1",
           "frames": T, "objects": N, "candidates": a.candidates, "refinement": results,
           "node_level_cap_frames_per_s": {"one_video": best, "eight_videos_one_per_gpu": round(8 * best, 1)},
           "producers_frames_per_s_8_gpus": "8 x ~53 (bench.py, fp32)",
           "reading": "the merge stage scales with the number of videos in flight (replicas only), never with GPUs per video: a "
                      "single long video is bounded by `one_video`; DAVIS-2017 val (30 videos) keeps all 8 GPUs busy"}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
