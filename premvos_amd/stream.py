#!/usr/bin/env python
"""Stages A (flow), B (proposals, general + specific weights), C (combine) and D (refinement) of simple_run.sh:21-58 as ONE
streaming process: every JPEG is decoded once (the four stage scripts of the reference decode it four times), the three
GPU stages run on their own host threads and HIP streams connected by bounded queues, files are written by a background
thread.  The output tree is the one the stage drivers write -- byte for byte (tests/test_gpu_plumbing.py) -- so the unchanged
ReID and MergeTrack stages read it as before:

    output/intermediate/flow/<seq>/<frame>.flo                      (named by the first frame of the pair, none for the last)
    output/intermediate/{general,specific,combined}_proposals/<seq>/<frame>.json
    output/intermediate/refined_proposals/<seq>/<frame>.json

This is SURVEY 8(f) rank 4 (host / format fast paths); the per-stage drivers (premvos_amd.{flow,proposal,refinement}.driver)
remain the drop-in twins of the reference's scripts.  rocJPEG is not part of the image, so decoding stays on the host
(PIL / libjpeg-turbo) on PREMVOS_IO_THREADS threads.

    python -m premvos_amd.stream --root <PReMVOS root> [--batch 8] [weights as in tools/run_stages.py]
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import queue
import sys
import threading
from typing import List, Optional

import numpy as np
import torch

from . import io_pipeline as iop

_END = object()


def _stage_thread(name, fn, q_in, q_out, errors):
    def run():
        try:
            while True:
                item = q_in.get()
                if item is _END:
                    break
                out = fn(item)
                if q_out is not None:
                    q_out.put(out)
        except BaseException as e:              # noqa: BLE001 -- re-raised by the caller
            errors.append(e)
            while q_in.get() is not _END:       # keep the producer from blocking on a full queue
                pass
        finally:
            if q_out is not None:
                q_out.put(_END)
    t = threading.Thread(target=run, name=f"premvos-{name}", daemon=True)
    t.start()
    return t


def run(root: str, seq_file: str, flow_weights: str, general_weights: str, specific_weights: str, refinement_weights: str,
        batch: int = 8, image_dir: str = "data/DAVIS/JPEGImages/480p/", out: str = "output/intermediate") -> int:
    from PIL import Image

    from .flow import pwc_dc_net
    from .flow.driver import FlowStage, writeFlowFile
    from .proposal import driver as pd
    from .refinement import driver as rd

    os.chdir(root)
    dev = "cuda"
    flow_net = pwc_dc_net(flow_weights).cuda().eval()
    nets = []
    for wfile in (general_weights, specific_weights):
        w = pd.load_weights(wfile)
        nets.append(pd.ProposalNet(w, num_blocks=pd.infer_num_blocks(w)))
    rw = rd.load_weights(refinement_weights)
    engine = rd.RefinementEngine(rd.RefinementNet(rw, rd.infer_num_middle(rw)))
    with open(seq_file) as f:
        folders = [ln.rstrip() for ln in f if ln.rstrip()]

    errors: List[BaseException] = []
    writer = iop.Writer(enabled=True)
    q_flow, q_prop, q_ref = (queue.Queue(maxsize=3) for _ in range(3))
    streams = {k: torch.cuda.Stream(device=dev) for k in ("flow", "prop", "ref")}
    flow_stages, prop_stages = {}, {}

    def dump_json(fn, obj):
        os.makedirs(os.path.dirname(fn), exist_ok=True)
        with open(fn, "w") as f:
            json.dump(obj, f)

    def flow_work(chunk):                           # chunk: (seq, names, frames [n,H,W,3] uint8 RGB, next frame or None)
        seq, names, frames, nxt = chunk
        second = list(frames[1:]) + ([nxt] if nxt is not None else [])
        n = len(second)                             # pairs in this chunk (the last frame of a video has none)
        if n == 0:
            return None
        with torch.cuda.stream(streams["flow"]):
            if n not in flow_stages:
                flow_stages[n] = FlowStage(net=flow_net, batch=n)
            im1 = torch.from_numpy(np.stack(frames[:n])).to(dev)
            im2 = torch.from_numpy(np.stack(second)).to(dev)
            flo = flow_stages[n].run(im1, im2).cpu().numpy()
        for k in range(n):
            fn = os.path.join(out, "flow", seq, names[k] + ".flo")
            os.makedirs(os.path.dirname(fn), exist_ok=True)
            writer.submit(writeFlowFile, fn, flo[k])
        return None

    def prop_work(chunk):
        seq, names, frames, _ = chunk
        n = len(frames)
        orig = frames[0].shape[:2]
        lists = []
        with torch.cuda.stream(streams["prop"]):
            x = torch.from_numpy(np.stack(frames)).to(dev)
            for which, net in enumerate(nets):
                key = (which, n)
                if key not in prop_stages:
                    prop_stages[key] = pd.ProposalStage({}, batch=n, device=net.device, net=net, rgb_input=True)
                st = prop_stages[key]
                st.run(x)
                lists.append([pd.convert_results_to_json(st.detections(k, orig)) for k in range(n)])
        combined = []
        for k in range(n):
            g, s = lists[0][k], lists[1][k]
            for sub, obj in (("general_proposals", g), ("specific_proposals", s), ("combined_proposals", g + s)):
                writer.submit(dump_json, os.path.join(out, sub, seq, names[k] + ".json"), obj)
            combined.append([dict(p) for p in g + s])           # the refinement stage adds keys to its own copies
        return seq, names, frames, combined

    def ref_work(item):
        seq, names, frames, combined = item
        with torch.cuda.stream(streams["ref"]):
            G = max(1, int(os.environ.get("PREMVOS_DRIVER_BATCH", "4")))
            for s0 in range(0, len(frames), G):
                engine.refine_frames(frames[s0:s0 + G], combined[s0:s0 + G])
            streams["ref"].synchronize()
        for k in range(len(frames)):
            writer.submit(dump_json, os.path.join(out, "refined_proposals", seq, names[k] + ".json"), combined[k])
        return None

    threads = [_stage_thread("flow", flow_work, q_flow, None, errors), _stage_thread("prop", prop_work, q_prop, q_ref, errors),
               _stage_thread("ref", ref_work, q_ref, None, errors)]
    n_frames = 0
    for video in folders:
        images = sorted(glob.glob(os.path.join(video, "*")))
        seq = video.rstrip("/").split("/")[-1]
        decoded = iop.prefetch(images, lambda fn: np.ascontiguousarray(np.asarray(Image.open(fn).convert("RGB"))[:, :, :3]))
        names = [os.path.splitext(os.path.basename(fn))[0] for fn in images]
        cur: List[np.ndarray] = []
        cur_names: List[str] = []
        pending = None                               # a full chunk waiting for the first frame of its successor

        def emit(chunk_frames, chunk_names, nxt):
            if nxt is not None and nxt.shape != chunk_frames[0].shape:
                nxt = None                           # (a size change inside a video: the reference would fail in cv2 here)
            item = (seq, chunk_names, chunk_frames, nxt)
            q_flow.put(item)
            q_prop.put(item)

        for name, fr in zip(names, decoded):
            if errors:
                break
            if pending is not None:
                emit(pending[0], pending[1], fr)
                pending = None
            if cur and fr.shape != cur[0].shape:
                emit(cur, cur_names, None)
                cur, cur_names = [], []
            cur.append(fr)
            cur_names.append(name)
            n_frames += 1
            if len(cur) == batch:
                pending, cur, cur_names = (cur, cur_names), [], []
        if pending is not None:
            emit(pending[0], pending[1], cur[0] if cur else None)
        if cur:
            emit(cur, cur_names, None)
    q_flow.put(_END)
    q_prop.put(_END)
    for t in threads:
        t.join()
    writer.close()
    if errors:
        raise errors[0]
    return n_frames


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default=".")
    ap.add_argument("--seq_file", default="seq_to_run.txt")
    ap.add_argument("--flow_weights", default="weights/PReMVOS_weights/optical_flow_net/pwc_net.pth.tar")
    ap.add_argument("--general_weights", default="weights/PReMVOS_weights/proposal_net/general_weights/proposal_general_weights")
    ap.add_argument("--specific_weights", default="weights/PReMVOS_weights/proposal_net/specific_weights/proposal_specific_weights")
    ap.add_argument("--refinement_weights", default="weights/PReMVOS_weights/refinement_net/specific_weights/refinement_specific_weights")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PREMVOS_STREAM_BATCH", "8")), help="frames per chunk")
    a = ap.parse_args(argv)
    n = run(a.root, a.seq_file, a.flow_weights, a.general_weights, a.specific_weights, a.refinement_weights, a.batch)
    print("frames:", n)
    return 0


if __name__ == "__main__":
    sys.exit(main())
