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
from . import jpeg

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


class StreamPipeline:
    """Nets, per-batch-size stage objects and HIP streams, built once; ``run_sequences`` may be called repeatedly."""

    def __init__(self, flow_weights: str, general_weights: str, specific_weights: str, refinement_weights: str,
                 batch: int = 8, out: str = "output/intermediate"):
        from .flow import pwc_dc_net
        from .proposal import driver as pd
        from .refinement import driver as rd
        self.batch, self.out, self.dev = batch, out, "cuda"
        # Launch lists are issued eagerly here (no HIP-graph capture / replay): four host threads drive the GPU at once, and a
        # capture on one thread while others launch is the one construct this driver avoids (an intermittent hang was seen
        # with it); the ~400 launches of a net cost ~2 ms of host time per call, hidden behind the other threads' work.
        self.flow_net = pwc_dc_net(flow_weights, use_graph=False).cuda().eval()
        self.nets = []
        for wfile in (general_weights, specific_weights):
            w = pd.load_weights(wfile)
            self.nets.append(pd.ProposalNet(w, num_blocks=pd.infer_num_blocks(w), use_graph=False))
        rw = rd.load_weights(refinement_weights)
        self.engine = rd.RefinementEngine(rd.RefinementNet(rw, rd.infer_num_middle(rw), use_graph=False))
        self.streams = {k: torch.cuda.Stream(device=self.dev) for k in ("flow", "prop0", "prop1", "ref", "decode")}
        self.flow_stages, self.prop_stages = {}, {}

    # ---- the stage bodies (each runs on its own host thread and HIP stream) ----------------------------------------
    def _flow(self, chunk, writer):                 # chunk: (seq, names, frames [n,H,W,3] uint8 RGB, next frame or None)
        from .flow.driver import FlowStage, writeFlowFile
        seq, names, frames, nxt = chunk
        second = list(frames[1:]) + ([nxt] if nxt is not None else [])
        n = len(second)                             # pairs in this chunk (the last frame of a video has none)
        if n == 0:
            return None
        with torch.cuda.stream(self.streams["flow"]):
            if n not in self.flow_stages:
                self.flow_stages[n] = FlowStage(net=self.flow_net, batch=n, use_graph=False)
            im1 = jpeg.stack_frames(frames[:n], self.dev)
            im2 = jpeg.stack_frames(second, self.dev)
            flo = self.flow_stages[n].run(im1, im2).cpu().numpy()
        for k in range(n):
            fn = os.path.join(self.out, "flow", seq, names[k] + ".flo")
            os.makedirs(os.path.dirname(fn), exist_ok=True)
            writer.submit(writeFlowFile, fn, flo[k])
        return None

    def _proposals(self, which, chunk, writer):
        from .proposal import driver as pd
        seq, names, frames, _ = chunk
        n, orig, net = len(frames), frames[0].shape[:2], self.nets[which]
        with torch.cuda.stream(self.streams[f"prop{which}"]):
            key = (which, n)
            if key not in self.prop_stages:
                self.prop_stages[key] = pd.ProposalStage({}, batch=n, device=net.device, net=net, rgb_input=True, use_graph=False)
            st = self.prop_stages[key]
            st.run(jpeg.stack_frames(frames, self.dev))
            lists = st.json_results(orig)
        sub = ("general_proposals", "specific_proposals")[which]
        for k in range(n):
            writer.submit(_dump_json, os.path.join(self.out, sub, seq, names[k] + ".json"), lists[k])
        return chunk, lists

    def _refine(self, item, writer):
        (seq, names, frames, _), general, specific = item
        combined = []
        for k in range(len(frames)):
            both = general[k] + specific[k]           # combine_general_and_specific.py:33
            writer.submit(_dump_json, os.path.join(self.out, "combined_proposals", seq, names[k] + ".json"), both)
            combined.append([dict(p) for p in both])  # the refinement stage adds keys to its own copies
        with torch.cuda.stream(self.streams["ref"]):
            G = max(1, int(os.environ.get("PREMVOS_DRIVER_BATCH", "4")))
            for s0 in range(0, len(frames), G):
                self.engine.refine_frames(frames[s0:s0 + G], combined[s0:s0 + G])
            self.streams["ref"].synchronize()
        for k in range(len(frames)):
            writer.submit(_dump_json, os.path.join(self.out, "refined_proposals", seq, names[k] + ".json"), combined[k])
        return None

    # ---- the driver ---------------------------------------------------------------------------------------------------
    def run_sequences(self, folders: List[str]) -> int:
        errors: List[BaseException] = []
        writer = iop.Writer(enabled=True)
        q_flow, q_g, q_s, q_rg, q_rs = (queue.Queue(maxsize=3) for _ in range(5))
        q_join: "queue.Queue" = queue.Queue(maxsize=3)

        def join_props():                            # pairs the general and the specific result of the same chunk
            try:
                while True:
                    a, b = q_rg.get(), q_rs.get()
                    if a is _END or b is _END:
                        for item, q in ((a, q_rg), (b, q_rs)):      # one side failed early: let the other run out
                            while item is not _END:
                                item = q.get()
                        break
                    q_join.put((a[0], a[1], b[1]))
            finally:
                q_join.put(_END)
        joiner = threading.Thread(target=join_props, name="premvos-join", daemon=True)
        joiner.start()
        threads = [_stage_thread("flow", lambda c: self._flow(c, writer), q_flow, None, errors),
                   _stage_thread("prop-general", lambda c: self._proposals(0, c, writer), q_g, q_rg, errors),
                   _stage_thread("prop-specific", lambda c: self._proposals(1, c, writer), q_s, q_rs, errors),
                   _stage_thread("refine", lambda it: self._refine(it, writer), q_join, None, errors), joiner]
        n_frames = 0
        for video in folders:
            images = sorted(glob.glob(os.path.join(video, "*")))
            seq = video.rstrip("/").split("/")[-1]
            # PREMVOS_GPU_JPEG=1: the pool only Huffman-decodes; this thread finishes each frame ONCE on the GPU (inverse DCT,
            # up-sampling, colour conversion) and the four stage threads share the HBM copy instead of uploading it each
            decoded = iop.prefetch(images, jpeg.loader())
            names = [os.path.splitext(os.path.basename(fn))[0] for fn in images]
            cur: List[np.ndarray] = []
            cur_names: List[str] = []
            pending = None                           # a full chunk waiting for the first frame of its successor

            def emit(chunk_frames, chunk_names, nxt):
                if nxt is not None and nxt.shape != chunk_frames[0].shape:
                    nxt = None                       # (a size change inside a video: the reference would fail in cv2 here)
                item = (seq, chunk_names, chunk_frames, nxt)
                for q in (q_flow, q_g, q_s):
                    q.put(item)

            for name, fr in zip(names, decoded):
                if errors:
                    break
                if isinstance(fr, jpeg.Decoded):
                    with torch.cuda.stream(self.streams["decode"]):
                        fr = jpeg.to_device(fr, self.dev)
                    self.streams["decode"].synchronize()
                if pending is not None:
                    emit(pending[0], pending[1], fr)
                    pending = None
                if cur and fr.shape != cur[0].shape:
                    emit(cur, cur_names, None)
                    cur, cur_names = [], []
                cur.append(fr)
                cur_names.append(name)
                n_frames += 1
                if len(cur) == self.batch:
                    pending, cur, cur_names = (cur, cur_names), [], []
            if pending is not None:
                emit(pending[0], pending[1], cur[0] if cur else None)
            if cur:
                emit(cur, cur_names, None)
        for q in (q_flow, q_g, q_s):
            q.put(_END)
        for t in threads:
            t.join()
        writer.close()
        if errors:
            raise errors[0]
        return n_frames


def _dump_json(fn, obj):
    os.makedirs(os.path.dirname(fn), exist_ok=True)
    with open(fn, "w") as f:
        json.dump(obj, f)


def run(root: str, seq_file: str, flow_weights: str, general_weights: str, specific_weights: str, refinement_weights: str,
        batch: int = 8, out: str = "output/intermediate") -> int:
    os.chdir(root)
    pipe = StreamPipeline(flow_weights, general_weights, specific_weights, refinement_weights, batch, out)
    with open(seq_file) as f:
        folders = [ln.rstrip() for ln in f if ln.rstrip()]
    return pipe.run_sequences(folders)


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
