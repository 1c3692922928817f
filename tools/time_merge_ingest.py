#!/usr/bin/env python
"""Can the merge rank of an 8-GPU job keep up?  (the one multi-GPU question a 1-GPU lease can answer.)

north_star: "frames shard across the 8 GPUs of one node, a single RCCL gather over xGMI to collect masks for the CPU-side merge".
In the product (`python -m premvos_amd.stream --gpus 8 --gather`) that gather ends in `stream.DeviceGather.decode_round` on rank 0:
every rank's packed buffer of a round of chunks becomes .flo / proposal / combined / refined-with-RLE files -- what the reference
does per rank through the file system (MergeTrack/merge.py:66-67,126-128 reads what FewShotSegmentationForwarder.py:137-155 wrote).
Rank 0 computes its own chunks at the same time.  At 8 ranks x ~53 frames/s the merge rank has to ingest ~430 frames/s.

This tool measures that on ONE GPU, without RCCL:

  1. reference : the streaming driver, one rank, per-rank writers            -> tree R (and the rank's own file-to-file rate)
  2. record    : the same frames with --gather semantics at world 1; every packed buffer the exchange sends is kept (HBM)
  3. alone     : 8 ranks' worth of rounds replayed into DeviceGather.decode_round + the writer threads, nothing else running
  4. beside    : a fake 8-rank job -- rank 0 runs the REAL StreamPipeline on its own copy of the clip (decode, four stages, pack,
                 exchange thread) and every round's gather "delivers" its own buffer + the recorded buffers of 7 other ranks
                 (device-resident, as RCCL leaves them); frames/s = 8 x clip / wall time
  every tree of 3 and 4 (clip0 .. clip7) must be byte-identical to R.

`--legacy` replays the round-5 form of the merge side for comparison (bit-unpack + RLE boundaries on the merge rank's GPU per frame,
pageable copies, json.dump, ONE writer thread).

    python tools/time_merge_ingest.py [--frames 64] [--world 8] [--chunk 8] [--legacy] [--out gpurun_out/merge_ingest.json]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def object_like_refinement_weights(shift: float = 7.5):
    """premvos_amd.synth.refinement_weights(0) with the foreground logit's bias raised: with the plain random weights every mask of
    the synthetic clip is EMPTY (fg - bg logit = -9 +- 2.6 over a crop: checked with oracle/refinement_oracle.py), so the RLE /
    string / JSON side of a file-to-file run had nothing to do.  +7.5 gives blob-shaped masks of 4 ... 22 k pixels and 1300 ... 2100
    run boundaries each -- on the heavy side of a DAVIS object mask.  Used for the file-to-file legs only; bench.py's `value` runs
    the unmodified synthetic weights (mask content does not change its work)."""
    from premvos_amd import synth
    w = synth.refinement_weights(0)
    w["logits/features/biases"] = w["logits/features/biases"] + torch.tensor([0.0, shift])
    return w


def build_job(root: str, n_frames: int, world: int, h: int = 480, w: int = 854, weights: bool = True) -> dict:
    """A synthetic 480p JPEG clip (quality 95), ``world`` names for it (clip0 = the files, clip1.. = links) and the four weight files."""
    from PIL import Image
    from premvos_amd import synth
    base = os.path.join(root, "data", "DAVIS", "JPEGImages", "480p")
    seq = os.path.join(base, "clip0")
    os.makedirs(seq)
    for s0 in range(0, n_frames, 32):
        fr = synth.clip_frames(s0, min(s0 + 32, n_frames), h, w).numpy()
        for i, im in enumerate(fr):
            Image.fromarray(im).save(os.path.join(seq, f"{s0 + i:05d}.jpg"), quality=95)
    for r in range(1, world):
        os.symlink(seq, os.path.join(base, f"clip{r}"))
    clips = [os.path.join(base, f"clip{r}") + "/" for r in range(world)]
    if not weights:
        return {"clips": clips, "weights": None}
    wd = os.path.join(root, "weights")
    os.makedirs(wd)
    torch.save({"state_dict": synth.pwc_state_dict(0)}, os.path.join(wd, "pwc.pth.tar"))
    torch.save(synth.proposal_weights(0), os.path.join(wd, "general.pt"))
    torch.save(synth.proposal_weights(1), os.path.join(wd, "specific.pt"))
    torch.save(object_like_refinement_weights(), os.path.join(wd, "refine.pt"))
    return {"clips": clips,
            "weights": [os.path.join(wd, n) for n in ("pwc.pth.tar", "general.pt", "specific.pt", "refine.pt")]}


def tree_digest(out: str) -> dict:
    """{stage/<file> -> digest} per sequence: {seq: {relative name without the sequence: blake2b}}."""
    per = {}
    for root, _, files in os.walk(out):
        for f in files:
            fn = os.path.join(root, f)
            stage, seq = os.path.relpath(root, out).split(os.sep)[:2]
            hsh = hashlib.blake2b(digest_size=16)
            with open(fn, "rb") as fh:
                while True:
                    blk = fh.read(1 << 22)
                    if not blk:
                        break
                    hsh.update(blk)
            per.setdefault(seq, {})[f"{stage}/{f}"] = hsh.hexdigest()
    return per


class _Timer:
    """Wraps a bound method: seconds spent inside it (on whatever thread calls it) and the number of calls."""

    def __init__(self, obj, name):
        self.s, self.n, self._fn, self.starts = 0.0, 0, getattr(obj, name), []
        setattr(obj, name, self)

    def __call__(self, *a, **k):
        t = time.perf_counter()
        self.starts.append(t)
        try:
            return self._fn(*a, **k)
        finally:
            self.s += time.perf_counter() - t
            self.n += 1


class _DepthSampler:
    def __init__(self, writer, period=0.005):
        self.w, self.period, self.samples, self._stop = writer, period, [], threading.Event()
        self.t = threading.Thread(target=self._loop, daemon=True)

    def _loop(self):
        while not self._stop.is_set():
            self.samples.append(self.w._q.qsize())
            self._stop.wait(self.period)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self.t.join()

    def summary(self):
        s = self.samples or [0]
        return {"mean": round(sum(s) / len(s), 1), "max": max(s), "capacity": self.w._q.maxsize}


def _legacy_decode_round(dg, k, bufs, out, writer):
    """The merge side as round 5 shipped it (stream.DeviceGather.decode_and_write of commit a3bf600), on today's buffer layout:
    every rank's masks bit-unpacked and their run boundaries found on the merge rank's GPU frame by frame (two synchronous copies
    each), pageable device-to-host copies of flow and the small arrays, json.dump, everything handed to ONE writer thread."""
    from premvos_amd.flow.driver import writeFlowFile
    from premvos_amd.mergetrack import encode_masks_begin, encode_masks_finish
    from premvos_amd.proposal.driver import custom_resize_shape, results_json

    def dump(fn, obj):
        os.makedirs(os.path.dirname(fn), exist_ok=True)
        with open(fn, "w") as f:
            json.dump(obj, f)

    def flo(fn, uv):
        os.makedirs(os.path.dirname(fn), exist_ok=True)
        writeFlowFile(fn, uv)
    files = 0
    for r in range(dg.world):
        if k >= len(dg.chunks[r]):
            continue
        seq, names, has_next, (h, w) = dg.chunks[r][k]
        n = len(names)
        u = dg.x.unpack(bufs[r])
        flow = u["flow"][:n, :h, :w].cpu().numpy()
        for i in range(n - (0 if has_next else 1)):
            writer.submit(flo, os.path.join(out, "flow", seq, names[i] + ".flo"), np.array(flow[i], copy=True, order="C"))
        nh, nw = custom_resize_shape(h, w)
        scale = (nh * 1.0 / h + nw * 1.0 / w) / 2
        lists = {}
        for which in ("general", "specific"):
            lists[which] = results_json(u[which + "_boxes"][:n].cpu().numpy(), u[which + "_probs"][:n].cpu().numpy(),
                                        u[which + "_count"][:n].cpu().numpy(), scale, (h, w))
            for i in range(n):
                writer.submit(dump, os.path.join(out, which + "_proposals", seq, names[i] + ".json"), lists[which][i])
        conf = u["conf"][:n].cpu().numpy().copy()
        masks = u["masks"]
        for i in range(n):
            both = lists["general"][i] + lists["specific"][i]
            writer.submit(dump, os.path.join(out, "combined_proposals", seq, names[i] + ".json"), both)
            refined = [dict(p) for p in both]
            handle = encode_masks_begin(masks[i, :len(both), :h, :w].contiguous()) if both else None

            def finish(refined=refined, handle=handle, c=conf[i], fn=os.path.join(out, "refined_proposals", seq, names[i] + ".json")):
                if handle is not None:
                    for q, seg, cv in zip(refined, encode_masks_finish(handle), c):
                        q["segmentation"] = seg
                        q["conf_score"] = str(cv)
                dump(fn, refined)
            writer.submit(finish)
        files += 5 * n - (0 if has_next else 1)
    return files


def measure(sp, clips, n_frames: int, world: int = 8, legacy: bool = False, modes=("alone", "beside"), tmp_out: str = None,
            reference: bool = True) -> dict:
    """``sp``: a premvos_amd.stream.StreamPipeline (plans may be warm or cold: step 1 / 2 warm them).  ``clips``: ``world`` folder
    names of ONE synthetic clip of ``n_frames`` frames (build_job).  Returns the report; raises if a tree differs from the reference."""
    from premvos_amd import io_pipeline as iop
    from premvos_amd import stream
    dev = torch.device("cuda", torch.cuda.current_device())
    B = sp.batch
    tmp = tmp_out or tempfile.mkdtemp(prefix="premvos_ingest_")
    saved_out = sp.out
    rep = {"frames_per_clip": n_frames, "chunk": B, "fake_ranks": world, "merge_side": "round-5 form" if legacy else "round-6 form",
           "what": "tools/time_merge_ingest.py: the merge rank of an 8-rank `premvos_amd.stream --gather` job on ONE GPU -- recorded packed "
                   "buffers of 7 ranks replayed into DeviceGather.decode_round + the writer threads (a) alone, (b) beside the real "
                   "StreamPipeline computing rank 0's own share; every written tree compared with the one-rank tree"}
    try:
        # 1. the one-rank tree (per-rank writers), twice: the first run builds the plans
        ref_out = os.path.join(tmp, "ref")
        sp.out = ref_out
        t_ref = []
        for _ in range(2 if reference else 1):
            shutil.rmtree(ref_out, ignore_errors=True)
            torch.cuda.synchronize()
            t = time.perf_counter()
            assert sp.run_sequences([clips[0]]) == n_frames
            torch.cuda.synchronize()
            t_ref.append(time.perf_counter() - t)
        ref = tree_digest(ref_out)["clip0"]
        rep["own_file_to_file_fps"] = round(n_frames / min(t_ref), 2)
        shutil.rmtree(ref_out, ignore_errors=True)

        # 2. record the buffers one rank sends (world 1: no process group; the merge side runs too and must reproduce the tree)
        rec_out = os.path.join(tmp, "rec")
        sp.out = rec_out
        plan1 = [[(0, 0, n_frames)]]
        dg1 = stream.DeviceGather(sp, clips[:1], plan1, 0, 1, dev)
        recorded = []
        send = dg1.x.exchange_async

        def recording(r):
            slot = send(r)
            recorded.append(dg1.x._packed[slot].clone())
            return slot
        dg1.x.exchange_async = recording
        torch.cuda.synchronize()
        t = time.perf_counter()
        assert sp.run_sequences(clips[:1], plan1[0], gather=dg1) == n_frames
        torch.cuda.synchronize()
        t_rec = time.perf_counter() - t
        assert len(recorded) == dg1.rounds == -(-n_frames // B)
        if tree_digest(rec_out)["clip0"] != ref:
            raise AssertionError("the gathered world-1 run does not reproduce the one-rank tree")
        shutil.rmtree(rec_out, ignore_errors=True)
        rep["own_gather_mode_fps"] = round(n_frames / t_rec, 2)
        rep["buffer_mb"] = round(dg1.x.nbytes / 1e6, 1)
        rep["host_prefix_mb"] = round(dg1.x.prefix_bytes / 1e6, 1)
        offs = [int(b[dg1.x.off_rle_off:dg1.x.off_rle_off + (dg1.x.n_masks + 1) * 4].view(torch.int32)[-1]) for b in recorded]
        rep["run_boundaries_per_chunk"] = {"mean": round(sum(offs) / len(offs)), "max": max(offs), "pool_capacity": dg1.x.pool_cap}

        plans = [[(r, 0, n_frames)] for r in range(world)]
        n_writers = 1 if legacy else iop.writer_threads(merge_rank=True)
        rep["writer_threads"] = n_writers

        def check(out, tag):
            got = tree_digest(out)
            if sorted(got) != [f"clip{r}" for r in range(world)]:
                raise AssertionError(f"{tag}: sequences written: {sorted(got)}")
            for seq, files in got.items():
                if files != ref:
                    bad = [k for k in ref if files.get(k) != ref[k]][:3]
                    raise AssertionError(f"{tag}: {seq} differs from the one-rank tree, e.g. {bad}")
            return sum(len(v) for v in got.values())

        # 3. ingest alone
        if "alone" in modes:
            out = os.path.join(tmp, "alone")
            dg = stream.DeviceGather(sp, clips, plans, 0, world, dev)
            dec = (lambda k, bufs, o, w_: _legacy_decode_round(dg, k, bufs, o, w_)) if legacy else dg.decode_round
            writer = iop.Writer(threads=n_writers)
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                torch.cuda.synchronize()
                with _DepthSampler(writer) as depth:
                    t = time.perf_counter()
                    t_dec = 0.0
                    for k in range(dg.rounds):
                        t1 = time.perf_counter()
                        dec(k, [recorded[k]] * world, out, writer)
                        t_dec += time.perf_counter() - t1
                    writer.close()
                    dt = time.perf_counter() - t
            files = check(out, "alone")
            shutil.rmtree(out, ignore_errors=True)
            rep["alone"] = {"frames_per_s": round(world * n_frames / dt, 1), "seconds": round(dt, 3), "files": files,
                            "exchange_thread_decode_ms_per_round": round(1e3 * t_dec / dg.rounds, 2),
                            "exchange_thread_busy_share": round(t_dec / dt, 3),
                            "writer_busy_share_per_thread": round(writer.busy_s / dt / n_writers, 3), "writer_queue_depth": depth.summary(),
                            "rle_overflow_chunks": dg.rle_overflows, "byte_identical_trees": True}

        # 4. ingest beside the rank's own compute: the real driver, the gather replaced by "own buffer + 7 recorded ones"
        if "beside" in modes:
            out = os.path.join(tmp, "beside")
            sp.out = out
            dg = stream.DeviceGather(sp, clips, plans, 0, world, dev)
            x, rounds = dg.x, {"k": 0}
            pack = x.pack

            def replayed_exchange(r):
                slot = x._n % len(x._packed)
                x._n += 1
                own = pack(r, slot)
                k = rounds["k"]
                rounds["k"] += 1
                x._gathered[slot] = [own] + [recorded[k % len(recorded)]] * (world - 1)
                return slot
            x.exchange_async = replayed_exchange
            if legacy:
                dg.decode_round = lambda k, bufs, o, w_: _legacy_decode_round(dg, k, bufs, o, w_)
            t_dec, t_xch = _Timer(dg, "decode_round"), _Timer(x, "exchange_async")
            writer = iop.Writer(threads=n_writers)
            torch.cuda.synchronize()
            with _DepthSampler(writer) as depth:
                t = time.perf_counter()
                assert sp.run_sequences(clips, plans[0], writer=writer, gather=dg) == n_frames
                writer.close()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t
            files = check(out, "beside")
            shutil.rmtree(out, ignore_errors=True)
            fps = world * n_frames / dt
            rep["beside"] = {"frames_per_s": round(fps, 1), "seconds": round(dt, 3), "files": files,
                             "own_chunks_fps": round(n_frames / dt, 2),
                             "vs_8x_own_gather_mode_rate": round(fps / (world * n_frames / t_rec), 3),
                             "exchange_thread_decode_ms_per_round": round(1e3 * t_dec.s / max(t_dec.n, 1), 2),
                             "exchange_thread_pack_ms_per_round": round(1e3 * t_xch.s / max(t_xch.n, 1), 2),
                             "exchange_thread_busy_share": round((t_dec.s + t_xch.s) / dt, 3),
                             "writer_busy_share_per_thread": round(writer.busy_s / dt / n_writers, 3), "writer_queue_depth": depth.summary(),
                             "rle_overflow_chunks": dg.rle_overflows, "byte_identical_trees": True,
                             "meets_430_frames_per_s": bool(fps >= 430.0),
                             "needed_for_8_ranks_at_own_rate": round(world * n_frames / t_rec, 1)}
        return rep
    finally:
        sp.out = saved_out
        if tmp_out is None:
            shutil.rmtree(tmp, ignore_errors=True)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64, help="frames of the clip (every fake rank owns one copy)")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("PREMVOS_STREAM_BATCH", "8")))
    ap.add_argument("--legacy", action="store_true", help="also replay the round-5 form of the merge side")
    ap.add_argument("--modes", default="alone,beside")
    ap.add_argument("--repeat", type=int, default=1, help="measure the round-6 form this many times (same pipeline object)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "merge_ingest.json"))
    a = ap.parse_args()
    from premvos_amd import stream
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(a.world))       # host threads per rank as on a full node (io_pipeline.host_budget)
    root = tempfile.mkdtemp(prefix="premvos_ingest_job_")
    try:
        job = build_job(root, a.frames, a.world)
        sp = stream.StreamPipeline(*job["weights"], batch=a.chunk, out=os.path.join(root, "output"))
        rep = {"now": measure(sp, job["clips"], a.frames, a.world, modes=tuple(a.modes.split(",")))}
        for i in range(1, a.repeat):
            rep[f"now_{i + 1}"] = measure(sp, job["clips"], a.frames, a.world, modes=tuple(a.modes.split(",")), reference=False)
        if a.legacy:
            rep["round5_form"] = measure(sp, job["clips"], a.frames, a.world, legacy=True, reference=False)
        rep["host"] = {"cpu_count": os.cpu_count(), "local_world_size_assumed": int(os.environ["LOCAL_WORLD_SIZE"])}
    finally:
        shutil.rmtree(root, ignore_errors=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))
    return 0


if __name__ == "__main__":
    sys.exit(main())
