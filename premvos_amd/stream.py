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

    python -m premvos_amd.stream --root <PReMVOS root> [--batch 8] [--gpus N [--gather]] [weights as in tools/run_stages.py]

``--gpus N``: one process per GPU (torch.distributed; RCCL), the videos -- or, with fewer videos than GPUs, chunk-aligned frame
ranges of each video -- shared out by premvos_amd.parallel.plan_shards; the output tree is byte-identical to the one-GPU run.
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

from . import _lib
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
    return iop.start_thread(run, f"premvos-{name}")        # (selects the creating thread's GPU first)


class StreamPipeline:
    """Nets, per-batch-size stage objects and HIP streams, built once; ``run_sequences`` may be called repeatedly."""

    def __init__(self, flow_weights: str, general_weights: str, specific_weights: str, refinement_weights: str,
                 batch: int = 8, out: str = "output/intermediate"):
        from .flow import pwc_dc_net
        from .proposal import driver as pd
        from .refinement import driver as rd
        self.batch, self.out, self.dev = batch, out, _lib.resolve_device()      # (this rank's GPU, indexed: the stage threads use THIS one)
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
        # refinement is half of a frame's FLOPs and has the longest host tail (D2H of masks' run boundaries): two lanes = two host
        # threads, each with its own stream and workspace of the net, take the chunks in turn (PREMVOS_STREAM_REFINE_LANES)
        # -- sized from the host threads this rank may use (io_pipeline.host_budget: cpu_count // ranks of the node)
        self.refine_lanes = max(1, int(os.environ.get("PREMVOS_STREAM_REFINE_LANES", iop.host_budget()["refine_lanes"])))
        self.streams = {k: torch.cuda.Stream(device=self.dev)
                        for k in ["flow", "prop0", "prop1", "decode"] + [f"ref{i}" for i in range(self.refine_lanes)]}
        self.flow_stages, self.prop_stages = {}, {}

    # ---- the stage bodies (each runs on its own host thread and HIP stream) ----------------------------------------
    def _flow(self, chunk, writer):                 # chunk: (seq, names, frames [n,H,W,3] uint8 RGB, next frame or None, staging)
        from .flow.driver import FlowStage
        seq, names, frames, nxt, stg = chunk
        second = list(frames[1:]) + ([nxt] if nxt is not None else [])
        n = len(second)                             # pairs in this chunk (the last frame of a video has none)
        if n == 0:
            if stg is not None:
                stg["_done"]("flow")
            return None
        with torch.cuda.stream(self.streams["flow"]):
            if n not in self.flow_stages:
                self.flow_stages[n] = FlowStage(net=self.flow_net, batch=n, use_graph=False)
            im1 = jpeg.stack_frames(frames[:n], self.dev)
            im2 = jpeg.stack_frames(second, self.dev)
            res = self.flow_stages[n].run(im1, im2)
            if stg is not None:                     # --gather: the result stays in HBM, in the chunk's staging block
                h, w = res.shape[1:3]
                stg["flow"][:n, :h, :w].copy_(res)
                self.streams["flow"].synchronize()
                stg["_done"]("flow")
                return None
            flo = res.cpu().numpy()
        for k in range(n):           # (the directory is made by whoever WRITES the file: with --gather that is the merge rank only)
            writer.submit(_write_flo, os.path.join(self.out, "flow", seq, names[k] + ".flo"), flo[k])
        return None

    def _proposals(self, which, chunk, writer):
        from .proposal import driver as pd
        seq, names, frames, _, stg = chunk
        n, orig, net = len(frames), frames[0].shape[:2], self.nets[which]
        with torch.cuda.stream(self.streams[f"prop{which}"]):
            key = (which, n)
            if key not in self.prop_stages:
                self.prop_stages[key] = pd.ProposalStage({}, batch=n, device=net.device, net=net, rgb_input=True, use_graph=False)
            st = self.prop_stages[key]
            st.run(jpeg.stack_frames(frames, self.dev))
            if stg is not None:                     # --gather: the detections go to the merge rank as the arrays they are
                tag, pl = ("general", "specific")[which], st.plan
                stg[tag + "_boxes"][:n].copy_(pl.final_boxes)
                stg[tag + "_probs"][:n].copy_(pl.final_probs)
                stg[tag + "_count"][:n].copy_(pl.final_count)
            lists = st.json_results(orig)           # (the refinement stage of THIS rank needs the rounded boxes; this also synchronises)
        if stg is not None:
            stg["_done"](("general", "specific")[which])
            return chunk, lists
        sub = ("general_proposals", "specific_proposals")[which]
        for k in range(n):
            writer.submit(_dump_json, os.path.join(self.out, sub, seq, names[k] + ".json"), lists[k])
        return chunk, lists

    def _refine(self, item, writer, lane: int = 0):
        (seq, names, frames, _, stg), general, specific = item
        if stg is not None:                           # --gather: masks + conf stay in HBM; the merge rank builds every JSON
            st = self.streams[f"ref{lane}"]
            with torch.cuda.stream(st):
                G = max(1, int(os.environ.get("PREMVOS_DRIVER_BATCH", "4")))
                for s0 in range(0, len(frames), G):
                    boxes = [[dict(p) for p in general[k] + specific[k]] for k in range(s0, min(s0 + G, len(frames)))]
                    self.engine.refine_frames_device(frames[s0:s0 + G], boxes, stg["masks"][s0:s0 + G], stg["conf"][s0:s0 + G], lane=lane)
                st.synchronize()
            stg["_done"]("refine")
            return None
        combined = []
        for k in range(len(frames)):
            both = general[k] + specific[k]           # combine_general_and_specific.py:33
            writer.submit(_dump_json, os.path.join(self.out, "combined_proposals", seq, names[k] + ".json"), both)
            combined.append([dict(p) for p in both])  # the refinement stage adds keys to its own copies
        # The GPU half runs here (this lane's stream and workspace); the host half -- run-length differencing and the ASCII packing
        # of the COCO "counts" strings, ~25 masks per frame -- is handed to the writer thread together with the JSON dump, so
        # this thread goes straight on to the next group's launches (round 2 packed the strings here: the thread was saturated).
        defer = getattr(writer, "runs_callables", False)
        finishers = []
        st = self.streams[f"ref{lane}"]
        with torch.cuda.stream(st):
            G = max(1, int(os.environ.get("PREMVOS_DRIVER_BATCH", "4")))
            for s0 in range(0, len(frames), G):
                finishers.append(self.engine.refine_frames(frames[s0:s0 + G], combined[s0:s0 + G], lane=lane, defer=defer))
            st.synchronize()
        paths = [os.path.join(self.out, "refined_proposals", seq, names[k] + ".json") for k in range(len(frames))]
        if defer:
            def finish_and_dump():
                for f in finishers:
                    if f is not None:
                        f()
                for fn, props in zip(paths, combined):
                    _dump_json(fn, props)
            writer.submit(finish_and_dump)
        else:
            for fn, props in zip(paths, combined):
                writer.submit(_dump_json, fn, props)
        return None

    def _decode_round(self, gather: "DeviceGather", prev, writer):
        """Merge rank: wait for the gather of round ``prev`` and turn every rank's buffer into files (other ranks: just wait)."""
        k, slot = prev
        gather.x.wait(slot)
        if gather.rank != gather.x.dst:
            return
        gather.decode_round(k, gather.x.gathered_slot(slot), self.out, writer)

    # ---- the driver ---------------------------------------------------------------------------------------------------
    def run_sequences(self, folders: List[str], shards: Optional[List[tuple]] = None, writer=None, gather: "Optional[DeviceGather]" = None) -> int:
        """``shards``: (index into folders, first frame, end frame) items (premvos_amd.parallel.plan_shards); None = every
        frame of every folder.  ``gather``: hand the results to the merge rank (``DeviceGather``) instead of writing them.
        Returns the number of frames this process owned."""
        errors: List[BaseException] = []
        own_writer = writer is None
        if own_writer:                               # (the merge rank of a gathered job writes every rank's files: several threads)
            merge = gather is not None and gather.world > 1 and gather.rank == gather.x.dst
            writer = iop.Writer(enabled=True, threads=iop.writer_threads(merge_rank=True) if merge else None)
        ready: "queue.Queue" = queue.Queue()          # --gather: chunks whose four parts are in their staging block, as they complete
        free_stg: "queue.Queue" = queue.Queue()
        xthread = None
        if gather is not None:
            for _ in range(4):                        # staging blocks in flight: decode-ahead + the stages + the exchange
                free_stg.put(gather.staging())

            def exchange_loop():
                """ONE thread issues the collectives, in chunk order: round k = this rank's k-th chunk (a filler when it has none);
                the gather of round k is in flight while round k + 1 is computed / packed; the merge rank decodes round k - 1."""
                try:
                    # the current device is per host thread and a new thread starts on device 0: this rank's GPU is selected before
                    # anything allocates, launches or synchronises here (io_pipeline.start_thread does it for every thread of the
                    # package; the explicit call keeps the exchange right even when the gather's device is not the creator's)
                    if gather.device.type == "cuda":
                        torch.cuda.set_device(gather.device)
                    x, pending, held = gather.x, {}, None
                    filler = gather.staging()
                    prev = None                        # (round, slot) whose gather was issued last
                    for k in range(gather.rounds):
                        if k < len(gather.chunks[gather.rank]):
                            while k not in pending:
                                idx, st_ = ready.get()
                                if idx is None:
                                    raise RuntimeError("a stage failed before its chunk was complete")
                                pending[idx] = st_
                            cur = pending.pop(k)
                        else:
                            cur = filler
                        slot = x.exchange_async(cur)   # pack (device) + the one gather of this round, not waited for
                        if cur is not filler:          # (packed: the staging block can take the next chunk)
                            for key in ("flow", "masks"):
                                cur[key].zero_()
                            if gather.device.type == "cuda":                 # packed and cleared before a stage stream writes it again
                                torch.cuda.current_stream(gather.device).synchronize()
                            free_stg.put(cur)
                        if prev is not None:
                            self._decode_round(gather, prev, writer)
                        prev = (k, slot)
                    if prev is not None:
                        self._decode_round(gather, prev, writer)
                except BaseException as e:             # noqa: BLE001
                    errors.append(e)
                    xdead.set()                        # the producer polls this: nobody returns staging blocks any more
            xdead = threading.Event()
            xthread = iop.start_thread(exchange_loop, "premvos-exchange")
        q_flow, q_g, q_s, q_rg, q_rs = (queue.Queue(maxsize=3) for _ in range(5))
        q_join: "queue.Queue" = queue.Queue(maxsize=3 + self.refine_lanes)

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
                for _ in range(self.refine_lanes):   # every consumer of q_join needs its own end marker (a single one left the
                    q_join.put(_END)                 # second refinement thread of a round-2 experiment blocked in get() for ever)
        joiner = iop.start_thread(join_props, "premvos-join")
        threads = [_stage_thread("flow", lambda c: self._flow(c, writer), q_flow, None, errors),
                   _stage_thread("prop-general", lambda c: self._proposals(0, c, writer), q_g, q_rg, errors),
                   _stage_thread("prop-specific", lambda c: self._proposals(1, c, writer), q_s, q_rs, errors)]
        threads += [_stage_thread(f"refine{i}", lambda it, i=i: self._refine(it, writer, i), q_join, None, errors)
                    for i in range(self.refine_lanes)] + [joiner]
        n_frames = n_chunks = 0
        if shards is None:
            shards = [(v, 0, None) for v in range(len(folders))]

        def finish(fr):
            # PREMVOS_GPU_JPEG=1: the pool only Huffman-decodes; this thread finishes each frame ONCE on the GPU (inverse DCT,
            # up-sampling, colour conversion) and the four stage threads share the HBM copy instead of uploading it each
            if isinstance(fr, jpeg.Decoded):
                with torch.cuda.stream(self.streams["decode"]):
                    fr = jpeg.to_device(fr, self.dev)
                self.streams["decode"].synchronize()
            return fr

        try:
            for v, first, end in shards:
                video = folders[v]
                images = sorted(glob.glob(os.path.join(video, "*")))
                seq = video.rstrip("/").split("/")[-1]
                # JPEG decode: Huffman pass on the decode-ahead pool, inverse DCT / up-sampling / colour conversion on the GPU, ONE
                # upload of coefficients per frame shared by the four stages (byte-identical to libjpeg-turbo; measured 46.0 ->
                # 49.2 frames/s file to file; PREMVOS_GPU_JPEG=0 = the library reader on the pool threads)
                for names, frames, nxt in iter_chunks(images, first, end, self.batch, jpeg.loader("1"), finish):
                    if errors:
                        break
                    stg = None
                    if gather is not None:
                        if n_chunks >= len(gather.chunks[gather.rank]):
                            raise RuntimeError(f"--gather: {seq} yields more chunks than the shard plan lists for rank {gather.rank} "
                                               f"({len(gather.chunks[gather.rank])}): every frame of a video must have one size")
                        if tuple(frames[0].shape[:2]) != tuple(gather.chunks[gather.rank][n_chunks][3]):
                            raise RuntimeError(f"--gather: {seq}/{names[0]} is {tuple(frames[0].shape[:2])}, the video's first frame is "
                                               f"{tuple(gather.chunks[gather.rank][n_chunks][3])}: one size per video required")
                        while stg is None:             # only the exchange thread returns blocks: do not outwait its death
                            try:
                                stg = free_stg.get(timeout=0.5)
                            except queue.Empty:
                                if errors or xdead.is_set() or not xthread.is_alive():
                                    raise RuntimeError("the exchange thread stopped before this chunk got a staging block")
                        stg["hw"] = tuple(frames[0].shape[:2])          # (ResultExchange.pack: the RLE runs over the frames' own size)
                        left, lock, idx = {"flow", "general", "specific", "refine"}, threading.Lock(), n_chunks

                        def done(part, stg=stg, left=left, lock=lock, idx=idx):
                            with lock:
                                left.discard(part)
                                last = not left
                            if last:
                                ready.put((idx, stg))
                        stg["_done"] = done
                    item = (seq, names, frames, nxt, stg)
                    for q in (q_flow, q_g, q_s):
                        q.put(item)
                    n_frames += len(frames)
                    n_chunks += 1
                if errors:
                    break
        except BaseException as e:                   # a decode / upload error on this thread: the stage threads must still end
            errors.insert(0, e)
        finally:
            for q in (q_flow, q_g, q_s):
                q.put(_END)
            for t in threads:
                t.join()
            if xthread is not None:
                if errors:
                    ready.put((None, None))          # wake the exchange thread: it fails its round instead of waiting for ever
                xthread.join()
            if own_writer:
                try:
                    writer.close()
                except BaseException as e:           # noqa: BLE001 -- reported below unless a stage failed first
                    errors.append(e)
        if errors:
            raise errors[0]
        return n_frames


def iter_chunks(images: List[str], first: int, end: Optional[int], batch: int, load, finish=lambda fr: fr):
    """Host logic of one shard, free of GPU code (tests/test_cpu_parallel.py drives it with a fake loader): frames
    [first, end) of the sorted file list ``images`` as chunks ``(names, frames, next_frame)`` of at most ``batch`` frames of
    one size; ``next_frame`` is the first frame after the chunk -- the second image of the chunk's last flow pair
    (script_pwc_multi.py:100-102) -- or None at the end of the video or in front of a change of size.  A shard that stops
    before the video does (``end`` < len(images): another rank owns the rest) decodes frame ``end`` too, for that purpose
    only.  Every frame is decoded once, ahead of the consumer (io_pipeline.prefetch)."""
    n = len(images)
    end = n if end is None else min(end, n)
    if first >= end:
        return
    todo = images[first:min(end + 1, n)]
    names = [os.path.splitext(os.path.basename(fn))[0] for fn in todo]
    decoded = iop.prefetch(todo, load)
    cur, cur_names = [], []
    pending = None                               # a full chunk waiting for the first frame of its successor

    def chunk(frames, chunk_names, nxt):
        if nxt is not None and nxt.shape != frames[0].shape:
            nxt = None                           # (a size change inside a video: the reference would fail in cv2 here)
        return chunk_names, frames, nxt

    for k, (name, fr) in enumerate(zip(names, decoded)):
        fr = finish(fr)
        if pending is not None:
            yield chunk(pending[0], pending[1], fr)
            pending = None
        if first + k >= end:                     # the boundary frame of the next rank's range: read, never owned
            if cur:                              # (a range that is not a multiple of the batch: its last chunk is short)
                yield chunk(cur, cur_names, fr)
                cur, cur_names = [], []
            break
        if cur and fr.shape != cur[0].shape:
            yield chunk(cur, cur_names, None)
            cur, cur_names = [], []
        cur.append(fr)
        cur_names.append(name)
        if len(cur) == batch:
            pending, cur, cur_names = (cur, cur_names), [], []
    if pending is not None:
        yield chunk(pending[0], pending[1], cur[0] if cur else None)
    if cur:
        yield chunk(cur, cur_names, None)


_DIRS: set = set()


def _mkdir_for(fn):
    d = os.path.dirname(fn)
    if d not in _DIRS:                          # (one stat per directory, not per file; a tree removed under a live process -- the
        os.makedirs(d, exist_ok=True)           #  bench's repeated runs -- is handled by the retry in the two writers below)
        _DIRS.add(d)


def _dump_json(fn, obj):
    # json.dumps = the encoder's C core in one shot; json.dump(obj, f) walks the object in Python and writes token by token -- same
    # text (FewShotSegmentationForwarder.py:151-155 / train.py:519-522 use json.dump), ~8x the time for a 40-proposal frame
    text = json.dumps(obj)
    _mkdir_for(fn)
    try:
        f = open(fn, "w")
    except FileNotFoundError:
        _DIRS.discard(os.path.dirname(fn))
        _mkdir_for(fn)
        f = open(fn, "w")
    with f:
        f.write(text)


def _write_flo(fn, uv):
    from .flow.driver import write_flo_raw
    _mkdir_for(fn)
    try:
        write_flo_raw(fn, uv)
    except FileNotFoundError:
        _DIRS.discard(os.path.dirname(fn))
        _mkdir_for(fn)
        write_flo_raw(fn, uv)


def _write_bytes(fn, data: bytes):
    os.makedirs(os.path.dirname(fn), exist_ok=True)
    with open(fn, "wb") as f:
        f.write(data)


def list_chunks(images: List[str], first: int, end: Optional[int], batch: int) -> List[tuple]:
    """GPU- and decode-free twin of ``iter_chunks`` for a video whose frames all have one size (what ``--gather`` requires): the
    chunks of frames [first, end) as (names, has_next) -- every rank computes every rank's list from the shard plan alone, so
    the merge rank knows which frames a gathered buffer holds without any side channel."""
    n = len(images)
    end = n if end is None else min(end, n)
    out = []
    for c0 in range(first, end, batch):
        c1 = min(c0 + batch, end)
        out.append(([os.path.splitext(os.path.basename(fn))[0] for fn in images[c0:c1]], c1 < n))
    return out


class DeviceGather:
    """``--gather`` (north_star: "a single RCCL gather over xGMI to collect masks for the CPU-side merge"): a rank writes NO files.
    Per chunk its results stay in HBM -- flow [n,H,W,2], the detections of both proposal nets, the refined masks [n,40,H,W] and
    conf -- are packed on the device into ONE fixed-size buffer (premvos_amd.parallel.ResultExchange: masks bit-packed) and go to
    the merge rank in ONE asynchronous gather per round of chunks (double-buffered: the collective of round k runs while round
    k + 1 computes; ranks that own no chunk in a round send a filler).  The merge rank -- the one that will run MergeTrack --
    turns every rank's buffer into files: the same bytes the per-rank writers produce, because proposal JSON, conf strings and RLE
    are pure functions of the gathered arrays (tests/test_gpu_plumbing.py compares the trees).  Round 6: the run boundaries of the
    masks are found by the PRODUCING rank while it packs and travel in the buffer, so the merge rank's share per chunk is one
    device-to-host copy + host work on several writer threads (``decode_round``) -- at 8 ranks it ingests ~430 frames/s beside
    its own chunks (profiles/r06_merge_ingest.json; round 5 unpacked and re-encoded every rank's masks on the merge rank's GPU and
    fed ONE writer thread).  Round 3 gathered host-packed file bytes per shard item instead."""

    P = 40                       # combined proposals per frame: 20 general + 20 specific (config.py:123)

    def __init__(self, pipe: "StreamPipeline", folders: List[str], plans: List[List[tuple]], rank: int, world: int, device,
                 pack_bits=None, unpack_bits=None, rle_pool=None):
        from PIL import Image
        from .parallel import ResultExchange
        self.pipe, self.rank, self.world = pipe, rank, world
        self.chunks: List[List[tuple]] = []          # per rank: (seq, names, has_next, (H, W))
        sizes = {}
        for r in range(world):
            mine = []
            for v, first, end in plans[r]:
                images = sorted(glob.glob(os.path.join(folders[v], "*")))
                if v not in sizes:
                    with Image.open(images[0]) as im:
                        sizes[v] = (im.size[1], im.size[0])
                seq = folders[v].rstrip("/").split("/")[-1]
                mine += [(seq, names, nxt, sizes[v]) for names, nxt in list_chunks(images, first, end, pipe.batch)]
            self.chunks.append(mine)
        self.rounds = max([len(c) for c in self.chunks] + [0])
        hm = max([hw[0] for hw in sizes.values()] + [8])
        wm = max([hw[1] for hw in sizes.values()] + [8])
        self.hm, self.wm = hm, wm
        self.x = ResultExchange(pipe.batch, hm, wm, self.P, device, pack_bits=pack_bits, unpack_bits=unpack_bits, rle_pool=rle_pool)
        self.device = _lib.resolve_device(device)    # indexed on the constructing (main) thread: the exchange thread selects THIS device
        self._free: "queue.Queue" = queue.Queue()    # page-locked host buffers (one chunk's prefix each), returned by the writer threads
        self._n_host, self._host_lock = 0, threading.Lock()
        self.rle_overflows = 0                       # chunks whose run boundaries did not fit the pool (encoded from the masks instead)
        if rank == self.x.dst and world > 1 and self.device.type == "cuda":
            # the merge rank's page-locked buffers are made now, not under the first rounds (locking 29 MB takes ~10 ms: 18 of them
            # inside the first two rounds showed as a start-up dip of the ingest rate)
            for _ in range(2 * world + 2):
                self._free.put(torch.empty(self.x.prefix_bytes, dtype=torch.uint8).pin_memory())
            self._n_host = 2 * world + 2

    def staging(self) -> dict:
        """The ``r`` dict of one chunk (ResultExchange.pack's input), zero-filled, at the job's largest frame size."""
        B, dev = self.pipe.batch, self.device
        return {"flow": torch.zeros((B, self.hm, self.wm, 2), dtype=torch.float32, device=dev),
                "masks": torch.zeros((B, self.P, self.hm, self.wm), dtype=torch.uint8, device=dev),
                "conf": torch.zeros((B, self.P), dtype=torch.float32, device=dev),
                "general_boxes": torch.zeros((B, 20, 4), dtype=torch.float32, device=dev),
                "general_probs": torch.zeros((B, 20), dtype=torch.float32, device=dev),
                "general_count": torch.zeros((B,), dtype=torch.int32, device=dev),
                "specific_boxes": torch.zeros((B, 20, 4), dtype=torch.float32, device=dev),
                "specific_probs": torch.zeros((B, 20), dtype=torch.float32, device=dev),
                "specific_count": torch.zeros((B,), dtype=torch.int32, device=dev)}

    # ---- the merge rank's side ------------------------------------------------------------------------------------------
    def _take_host(self, writer) -> torch.Tensor:
        """A page-locked host buffer for one chunk's prefix (flow, detections, conf, run boundaries): from the pool the writer
        threads return them to (filled at start-up on the merge rank), or a new one while fewer than 2 x world + 2 exist (a round
        in flight + a round being written)."""
        while True:
            try:
                return self._free.get_nowait()
            except queue.Empty:
                pass
            with self._host_lock:
                grow = self._n_host < 2 * self.world + 2
                if grow:
                    self._n_host += 1
            if grow:
                t = torch.empty(self.x.prefix_bytes, dtype=torch.uint8)
                return t.pin_memory() if self.device.type == "cuda" else t
            try:
                return self._free.get(timeout=0.5)
            except queue.Empty:                       # (only the writer threads return buffers: do not outwait their failure)
                if getattr(writer, "_err", None) is not None:
                    raise RuntimeError("the file writer failed while chunks were waiting for host buffers") from writer._err

    def decode_round(self, k: int, bufs, out: str, writer) -> int:
        """Merge rank: every rank's buffer of round ``k`` -> files.  Returns the number of files submitted.

        What this rank's GPU and this (exchange) thread do per chunk: ONE device-to-host copy of the buffer's prefix into page-locked
        memory (the copies of a round are queued back to back, one synchronisation) -- the run boundaries of the masks were found
        by the producing rank.  Everything else is host work per FRAME on the writer threads (``_finish_frame``): the .flo straight
        from the page-locked buffer, the proposal JSON from the detection arrays, the RLE strings from the run boundaries.  The
        same bytes the per-rank writers produce (tests/test_gpu_plumbing.py compares the trees; MergeTrack/merge.py:66-67,126-128
        and FewShotSegmentationForwarder.py:137-155 are what this hand-over replaces)."""
        live = [r for r in range(self.world) if k < len(self.chunks[r])]
        host = {}
        for r in live:
            host[r] = self._take_host(writer)
            host[r].copy_(bufs[r][:self.x.prefix_bytes], non_blocking=True)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        return sum(self.decode_and_write(r, k, bufs[r], out, writer, host[r]) for r in live)

    def decode_and_write(self, r: int, k: int, buf: torch.Tensor, out: str, writer, host: Optional[torch.Tensor] = None) -> int:
        """Merge rank: rank ``r``'s buffer of round ``k`` -> its files (``host``: the prefix, already on the host).  Returns the
        number of files submitted."""
        from .mergetrack import encode_masks_begin
        from .proposal.driver import custom_resize_shape
        seq, names, has_next, (h, w) = self.chunks[r][k]
        n = len(names)
        if host is None:
            host = self._take_host(writer)
            host.copy_(buf[:self.x.prefix_bytes])
        u = self.x.unpack(host, masks=False)
        view = {key: t.numpy() for key, t in u.items()}
        nh, nw = custom_resize_shape(h, w)
        scale = (nh * 1.0 / h + nw * 1.0 / w) / 2
        handles = [None] * n
        if int(view["rle_offsets"][-1]) > self.x.pool_cap:
            # more run boundaries than the pool holds (very ragged masks): this chunk is encoded from its masks, here
            self.rle_overflows += 1
            dev = buf if buf.is_cuda else buf.to(self.device)
            masks = self.x.unpack(dev)["masks"]
            for i in range(n):
                cnt = int(view["general_count"][i]) + int(view["specific_count"][i])
                if cnt:
                    handles[i] = encode_masks_begin(masks[i, :cnt, :h, :w].contiguous())
        lease = _Lease(host, self._free, n)
        for i in range(n):
            writer.submit(self._finish_frame, lease, view, i, seq, names[i], h, w, scale, has_next or i < n - 1, out, handles[i])
        return 5 * n - (0 if has_next else 1)

    def _finish_frame(self, lease, view, i, seq, name, h, w, scale, write_flow, out, handle):
        """Writer thread: the five files of one frame from the host copy of its chunk's buffer -- ONE call into libpremvos_hip.so
        (premvos_write_frame_files_host: number formatting, RLE strings, JSON text, file system calls, all without the interpreter
        lock, so the writer threads neither queue behind each other nor slow this rank's own launch threads).  The Python writers
        below (``_finish_frame_py``) produce the same bytes; they serve a chunk whose run boundaries overflowed the pool
        (``handle``) and PREMVOS_HOST_FILES=py (tests compare the two)."""
        try:
            if handle is not None or os.environ.get("PREMVOS_HOST_FILES", "c") == "py":
                return self._finish_frame_py(view, i, seq, name, h, w, scale, write_flow, out, handle)
            import ctypes as C
            f = _lib.FrameFiles()
            paths = [os.path.join(out, sub, seq, name + ext) for sub, ext in
                     (("flow", ".flo"), ("general_proposals", ".json"), ("specific_proposals", ".json"), ("combined_proposals", ".json"),
                      ("refined_proposals", ".json"))]
            flow = view["flow"]
            f.flo_path = paths[0].encode() if write_flow else None
            f.flow, f.flow_row_stride = flow[i].ctypes.data, flow.strides[1] // 4
            f.h, f.w, f.scale = h, w, float(np.float32(scale))
            gc, sc = int(view["general_count"][i]), int(view["specific_count"][i])
            f.boxes[0], f.boxes[1] = view["general_boxes"][i].ctypes.data, view["specific_boxes"][i].ctypes.data
            f.probs[0], f.probs[1] = view["general_probs"][i].ctypes.data, view["specific_probs"][i].ctypes.data
            f.count[0], f.count[1] = gc, sc
            for k in range(4):
                f.json_path[k] = paths[k + 1].encode()
            f.conf = view["conf"][i].ctypes.data
            f.rle_pool = view["rle_pool"].ctypes.data
            f.rle_offsets = view["rle_offsets"][i * self.P:].ctypes.data
            lib = _lib.load()
            rc = lib.premvos_write_frame_files_host(C.byref(f))
            if rc == 1:                                     # a directory is missing: make them all, once, and write again
                for fn in paths:
                    os.makedirs(os.path.dirname(fn), exist_ok=True)
                rc = lib.premvos_write_frame_files_host(C.byref(f))
            _lib.check(rc, "write_frame_files")
        finally:
            lease.done()

    def _finish_frame_py(self, view, i, seq, name, h, w, scale, write_flow, out, handle):
        from . import rle
        from .mergetrack import encode_masks_finish
        from .proposal.driver import results_json
        if write_flow:
            _write_flo(os.path.join(out, "flow", seq, name + ".flo"), view["flow"][i, :h, :w])
        lists = {}
        for which in ("general", "specific"):
            lists[which] = results_json(view[which + "_boxes"][i:i + 1], view[which + "_probs"][i:i + 1],
                                        view[which + "_count"][i:i + 1], scale, (h, w))[0]
            _dump_json(os.path.join(out, which + "_proposals", seq, name + ".json"), lists[which])
        both = lists["general"] + lists["specific"]
        _dump_json(os.path.join(out, "combined_proposals", seq, name + ".json"), both)
        refined = [dict(p) for p in both]
        if both:
            if handle is not None:
                segs = encode_masks_finish(handle)
            else:
                first = i * self.P
                strings = rle.strings_from_pool(view["rle_pool"], view["rle_offsets"][first:first + len(both) + 1], h * w)
                segs = [{"size": [h, w], "counts": c} for c in strings]
            for q, seg, cv in zip(refined, segs, view["conf"][i]):
                q["segmentation"] = seg
                q["conf_score"] = str(cv)
        _dump_json(os.path.join(out, "refined_proposals", seq, name + ".json"), refined)


class _Lease:
    """A host buffer shared by the per-frame writer calls of one chunk; the last one returns it to the pool."""

    def __init__(self, buf, pool: "queue.Queue", parts: int):
        self.buf, self._pool, self._left, self._lock = buf, pool, parts, threading.Lock()
        if parts <= 0:
            pool.put(buf)

    def done(self):
        with self._lock:
            self._left -= 1
            last = self._left == 0
        if last:
            self._pool.put(self.buf)


def _self_launch(gpus: int, argv: List[str]) -> int:
    """``python -m premvos_amd.stream --gpus N`` outside torch.distributed: start N ranks, one process per GPU."""
    import socket
    import subprocess
    backend = os.environ.get("PREMVOS_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < gpus and backend == "nccl":
        raise SystemExit(f"premvos_amd.stream --gpus {gpus}: this node exposes {ndev} GPU(s) (RCCL needs one device per rank)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "premvos_amd.stream"] + list(argv)
    return subprocess.call(cmd, env=env)


def run(root: str, seq_file: str, flow_weights: str, general_weights: str, specific_weights: str, refinement_weights: str,
        batch: int = 8, out: str = "output/intermediate", shard: str = "balanced", gather: bool = False,
        merge_share: float = 1.0) -> int:
    """One rank of the job (the only one when WORLD_SIZE is unset): device = LOCAL_RANK, work = its shards of the videos of
    ``seq_file`` (premvos_amd.parallel.plan_shards: whole videos when there are at least as many as ranks -- the reference's
    granularity, DAVISFewShotSegmentationDataset.py:130-150, merge.py:66-67,126-128 -- else chunk-aligned frame ranges with the
    boundary frame read as the second image of the last pair, script_pwc_multi.py:100-102).  No data-path collective; every
    rank writes its own files into the shared tree (the reference's filesystem rendezvous) unless ``gather``.  Returns the
    frames of the whole job."""
    import torch.distributed as dist
    from . import ops
    from .parallel import plan_shards
    os.chdir(root)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("PREMVOS_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit(f"{world} ranks but {ndev} GPU(s) visible: one device per rank is required")
    torch.cuda.set_device(local % max(ndev, 1))       # "cuda" below = this rank's device (gloo: ranks may share one, for tests)
    if world > 1 and not dist.is_initialized():
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=60))
    with open(seq_file) as f:
        folders = [ln.rstrip() for ln in f if ln.rstrip()]
    counts = [len(glob.glob(os.path.join(v, "*"))) for v in folders]
    # (--gather: rank 0 also writes every rank's files; --merge-share < 1 plans it as a slower rank, 0 = it computes nothing)
    share = merge_share if gather and world > 1 else 1.0
    plans = [plan_shards(counts, world, r, batch, shard, merge_share=share) for r in range(world)]
    pipe = StreamPipeline(flow_weights, general_weights, specific_weights, refinement_weights, batch, out)
    n = 0
    if gather and world > 1:
        # one round of the ONE gather per chunk (DeviceGather); every rank runs all rounds, with fillers where it owns nothing
        dg = DeviceGather(pipe, folders, plans, rank, world, pipe.dev)
        n = pipe.run_sequences(folders, plans[rank], gather=dg)
    else:
        n = pipe.run_sequences(folders, plans[rank])
    total = n
    if world > 1:
        t = torch.tensor([n], dtype=torch.int64, device=pipe.dev if backend == "nccl" else "cpu")
        dist.all_reduce(t)                                   # also the job's final barrier: every file is on disk after it
        total = int(t.item())
    if rank == 0:
        # which configurations computed these files (ops.tune_info): the shipped table's hash + how many signatures it lacked.
        # The manifest records the rank count and the shard plan, so it differs between a 1-rank and an N-rank job BY DESIGN: it
        # lives NEXT TO the stage tree (<out>/../premvos_amd_manifest.json), never inside `out` -- the tree the byte-identity
        # promise (and tests/test_gpu_plumbing.py::_same_tree) covers is `out` = output/intermediate, what ReID / MergeTrack read
        _dump_json(os.path.join(os.path.dirname(out.rstrip("/")) or ".", "premvos_amd_manifest.json"),
                   {"frames": total, "ranks": world, "chunk": batch, "sharding": shard, "merge_share": share,
                    "shards": [[[folders[v], a, b] for v, a, b in p] for p in plans], "conv_configurations": ops.tune_info()})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return total


def main(argv: Optional[List[str]] = None) -> int:
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default=".")
    ap.add_argument("--seq_file", default="seq_to_run.txt")
    ap.add_argument("--flow_weights", default="weights/PReMVOS_weights/optical_flow_net/pwc_net.pth.tar")
    ap.add_argument("--general_weights", default="weights/PReMVOS_weights/proposal_net/general_weights/proposal_general_weights")
    ap.add_argument("--specific_weights", default="weights/PReMVOS_weights/proposal_net/specific_weights/proposal_specific_weights")
    ap.add_argument("--refinement_weights", default="weights/PReMVOS_weights/refinement_net/specific_weights/refinement_specific_weights")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PREMVOS_STREAM_BATCH", "8")), help="frames per chunk")
    ap.add_argument("--gpus", type=int, default=int(os.environ.get("WORLD_SIZE", "1")),
                    help="ranks = GPUs of this node; > 1 outside torch.distributed.run starts the ranks itself")
    ap.add_argument("--shard", default="balanced", choices=["balanced", "contiguous"],
                    help="whole-video assignment when there are at least --gpus videos: by frame count, or the reference's slices")
    ap.add_argument("--gather", action="store_true",
                    help="hand every rank's results to rank 0 (ONE gather of one packed buffer per round of chunks); rank 0 writes every file")
    ap.add_argument("--merge-share", type=float, default=float(os.environ.get("PREMVOS_MERGE_SHARE", "1.0")),
                    help="with --gather: relative speed rank 0 is planned with (it also writes every rank's files: ~0.95 measured at 8 "
                         "ranks); 0 = rank 0 computes nothing, 1 = equal shares (default)")
    a = ap.parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _self_launch(a.gpus, argv)
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} was started with WORLD_SIZE={os.environ.get('WORLD_SIZE')}")
    n = run(a.root, a.seq_file, a.flow_weights, a.general_weights, a.specific_weights, a.refinement_weights, a.batch,
            shard=a.shard, gather=a.gather, merge_share=a.merge_share)
    if int(os.environ.get("RANK", "0")) == 0:
        print("frames:", n)
    return 0


if __name__ == "__main__":
    sys.exit(main())
