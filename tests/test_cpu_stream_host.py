"""Host logic of the streaming driver without a GPU: the thread / queue skeleton of StreamPipeline.run_sequences with fake stage
bodies -- normal completion over sharded ranges, and the shutdown paths (a stage error, a decode error on the
producer thread, two refinement consumers on one queue) must end every thread and re-raise the first error instead of hanging."""
import os
import threading
import time

import numpy as np
import pytest

from premvos_amd import stream


def _tree(tmp_path, n, bad=None):
    from PIL import Image
    d = tmp_path / "seq"
    d.mkdir()
    for t in range(n):
        fn = d / f"{t:05d}.png"
        if t == bad:
            fn.write_bytes(b"this is not an image")
        else:
            Image.fromarray(np.full((6, 8, 3), t, np.uint8)).save(fn)
    return str(d) + "/"


def _fake_pipeline(batch, lanes=2, fail_at=None, seen=None):
    p = object.__new__(stream.StreamPipeline)                   # no nets, no GPU: only what run_sequences touches
    p.batch, p.out, p.dev, p.refine_lanes, p.streams = batch, "out", "cpu", lanes, {}
    seen = seen if seen is not None else {"flow": [], "prop0": [], "prop1": [], "refine": [], "lanes": set()}
    lock = threading.Lock()

    def flow(chunk, writer):
        seq, names, frames, nxt, _stg = chunk
        ids = [int(f[0, 0, 0]) for f in frames] + ([int(nxt[0, 0, 0])] if nxt is not None else [])
        with lock:
            seen["flow"] += list(zip(ids[:-1], ids[1:]))

    def proposals(which, chunk, writer):
        ids = [int(f[0, 0, 0]) for f in chunk[2]]
        if fail_at is not None and which == 1 and fail_at in ids:
            raise RuntimeError(f"stage failure at frame {fail_at}")
        with lock:
            seen[f"prop{which}"] += ids
        return chunk, [[{"bbox": [0, 0, 1, 1], "score": 0.5}] for _ in ids]

    def refine(item, writer, lane=0):
        time.sleep(0.01)
        with lock:
            seen["refine"] += [int(f[0, 0, 0]) for f in item[0][2]]
            seen["lanes"].add(lane)

    p._flow, p._proposals, p._refine = flow, proposals, refine
    return p, seen


def _no_stage_threads_left():
    deadline = time.time() + 5
    while time.time() < deadline:
        if not [t for t in threading.enumerate() if t.name.startswith("premvos-")]:
            return True
        time.sleep(0.05)
    return False


def test_run_sequences_over_shards_feeds_every_stage_once(tmp_path, monkeypatch):
    monkeypatch.setenv("PREMVOS_GPU_JPEG", "0")
    folder = _tree(tmp_path, 11)
    p, seen = _fake_pipeline(batch=2)
    assert p.run_sequences([folder]) == 11
    assert sorted(seen["prop0"]) == sorted(seen["prop1"]) == sorted(seen["refine"]) == list(range(11))
    assert sorted(seen["flow"]) == [(t, t + 1) for t in range(10)] and seen["lanes"] == {0, 1}
    # two "ranks" over chunk-aligned ranges: together the same work, the boundary pair computed by the owner of its first frame
    a, sa = _fake_pipeline(batch=2)
    b, sb = _fake_pipeline(batch=2)
    assert a.run_sequences([folder], [(0, 0, 6)]) == 6 and b.run_sequences([folder], [(0, 6, 11)]) == 5
    assert sorted(sa["flow"] + sb["flow"]) == [(t, t + 1) for t in range(10)] and (5, 6) in sa["flow"]
    assert sorted(sa["refine"] + sb["refine"]) == list(range(11))
    assert _no_stage_threads_left()


@pytest.mark.parametrize("lanes", [1, 2, 3])
def test_a_stage_error_ends_every_thread_and_is_re_raised(tmp_path, monkeypatch, lanes):
    monkeypatch.setenv("PREMVOS_GPU_JPEG", "0")
    folder = _tree(tmp_path, 40)
    p, seen = _fake_pipeline(batch=2, lanes=lanes, fail_at=7)
    t0 = time.time()
    with pytest.raises(RuntimeError, match="stage failure at frame 7"):
        p.run_sequences([folder, folder])
    assert time.time() - t0 < 20 and _no_stage_threads_left()
    assert len(seen["prop0"]) < 80                                  # the producer stopped feeding soon after the failure


def test_a_decode_error_on_the_producer_thread_ends_every_thread(tmp_path, monkeypatch):
    monkeypatch.setenv("PREMVOS_GPU_JPEG", "0")
    folder = _tree(tmp_path, 12, bad=9)
    p, seen = _fake_pipeline(batch=4)
    with pytest.raises(Exception) as e:
        p.run_sequences([folder])
    assert "stage failure" not in str(e.value) and _no_stage_threads_left()
    assert sorted(seen["refine"]) == list(range(8))                 # the chunks in front of the damaged frame were handed over whole
                                                                    # (frame 8 had decoded: it is chunk [4,8)'s successor frame)

    class BadWriter:
        runs_callables = True

        def submit(self, fn, *a):
            pass

        def close(self):
            raise OSError("disk full")
    q, _ = _fake_pipeline(batch=4)
    (tmp_path / "b").mkdir()
    assert q.run_sequences([_tree(tmp_path / "b", 5)], writer=BadWriter()) == 5         # a caller-owned writer is closed by the caller


# ---------------------------------------------------------------------------------------------------------------------
# the merge rank's side of --gather, entirely on the host: a packed buffer -> the five files per frame
def _np_twins():
    import torch
    from test_cpu_parallel import _np_pack_bits, _np_rle_pool, _np_unpack_bits
    return {"pack_bits": _np_pack_bits, "unpack_bits": _np_unpack_bits, "rle_pool": _np_rle_pool}, torch


@pytest.mark.parametrize("writers,overflow,host_files,bigger", [(1, False, "c", False), (3, False, "c", False), (2, True, "c", False),
                                                                  (2, False, "py", False), (2, False, "c", True), (1, True, "py", True)])
def test_merge_rank_turns_gathered_buffers_into_the_per_rank_files(tmp_path, monkeypatch, writers, overflow, host_files, bigger):
    """DeviceGather.decode_round (round 6: run boundaries from the producing rank, per-frame host work on N writer threads, buffers
    leased from a pool) against a straightforward per-frame restatement of what a rank's own writers produce: flo_bytes of the flow
    window, results_json of the detections, rle.encode of every mask, str(conf).  ``overflow``: a pool too small for the chunk --
    the masks themselves are used (the GPU encoder is replaced by its numpy twin here).  ``host_files``: the interpreter-free C
    writer (premvos_write_frame_files_host, the default) or its Python twin -- both must produce the restatement's bytes.
    ``bigger``: another rank owns a video of LARGER frames, so the job's buffers are sized for those and this video's frames are
    the top-left window of every block (flow rows with a wider stride, run boundaries over the h x w window of a larger mask)."""
    import json as js
    monkeypatch.setenv("PREMVOS_HOST_FILES", host_files)
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    twins, torch = _np_twins()
    from premvos_amd import io_pipeline as iop
    from premvos_amd import mergetrack, rle
    from premvos_amd.flow.driver import flo_bytes
    from premvos_amd.proposal.driver import custom_resize_shape, results_json
    from PIL import Image
    h, w, B, T = 21, 30, 4, 7                      # two chunks: 4 + 3 frames; the second one ends the video (no .flo for its last frame)
    d = tmp_path / "JPEGImages" / "clip"
    d.mkdir(parents=True)
    for t in range(T):
        Image.fromarray(np.full((h, w, 3), t, np.uint8)).save(d / f"{t:05d}.png")
    pipe = type("P", (), {"batch": B})()
    monkeypatch.setenv("PREMVOS_GATHER_RLE_RUNS", "1" if overflow else "1024")
    if overflow:                                   # (the fallback encodes on the GPU: its numpy twin stands in)
        def begin(m):
            m = m.numpy()
            rows = [np.flatnonzero(np.diff(np.concatenate(([0], (x != 0).reshape(-1, order="F").astype(np.int8))))) for x in m]
            width = max([len(r) for r in rows] + [1])
            pos = np.zeros((len(rows), width), np.int32)
            for i, r in enumerate(rows):
                pos[i, :len(r)] = r
            return pos, np.array([len(r) for r in rows], np.int32), m.shape[1], m.shape[2]
        monkeypatch.setattr(mergetrack, "encode_masks_begin", begin)
    folders, plans = [str(d) + "/"], [[(0, 0, T)]]
    if bigger:
        d2 = tmp_path / "JPEGImages" / "wide"
        d2.mkdir()
        Image.fromarray(np.zeros((h + 5, w + 9, 3), np.uint8)).save(d2 / "00000.png")
        folders, plans = folders + [str(d2) + "/"], plans + [[(1, 0, 1)]]
    dg = stream.DeviceGather(pipe, folders, plans, 0, len(plans), "cpu", **twins)
    assert dg.rounds == 2 and dg.chunks[0][1][2] is False and (dg.hm, dg.wm) == ((h + 5, w + 9) if bigger else (h, w))
    rng = np.random.default_rng(3)
    out = str(tmp_path / "out")
    expect = {}
    with iop.Writer(threads=writers) as wr:
        for k, (seq, names, has_next, hw) in enumerate(dg.chunks[0]):
            n = len(names)
            r = dg.staging()
            r["hw"] = (h, w)
            r["flow"][:n, :h, :w] = torch.from_numpy(rng.standard_normal((n, h, w, 2)).astype(np.float32))
            gc, sc = rng.integers(0, 4, n), rng.integers(0, 4, n)
            r["general_count"][:n], r["specific_count"][:n] = torch.from_numpy(gc.astype(np.int32)), torch.from_numpy(sc.astype(np.int32))
            for key in ("general", "specific"):
                xy = rng.uniform(0, 40, (n, 20, 2)).astype(np.float32)
                r[key + "_boxes"][:n] = torch.from_numpy(np.concatenate([xy, xy + rng.uniform(1, 60, (n, 20, 2)).astype(np.float32)], -1))
                r[key + "_probs"][:n] = torch.from_numpy(rng.uniform(0.5, 1, (n, 20)).astype(np.float32))
            r["conf"][:n] = torch.from_numpy(rng.uniform(-1, 1, (n, dg.P)).astype(np.float32))
            r["masks"][:n, :, :h, :w] = torch.from_numpy((rng.random((n, dg.P, h, w)) > 0.55).astype(np.uint8))
            if bigger:                                 # (what lies outside the frame's window must not reach any file)
                r["masks"][:, :, h:, :] = 1
                r["masks"][:, :, :, w:] = 1
                r["flow"][:, h:] = 7.0
                r["flow"][:, :, w:] = 7.0
            slot = dg.x.exchange_async(r)
            dg.x.wait(slot)
            bufs = dg.x.gathered_slot(slot) * len(plans)              # (rank 1 owns one chunk in round 0: it gets the same bytes)
            files = dg.decode_round(k, bufs, out if not bigger else str(tmp_path / "out_all"), wr)
            if bigger:                                 # round 0 also decoded rank 1's (one-frame, last-of-video) chunk: 4 files
                assert files == 5 * n - (0 if has_next else 1) + (4 if k == 0 else 0)
                out_dir = str(tmp_path / "out_all")
            else:
                assert files == 5 * n - (0 if has_next else 1)
                out_dir = out
            # the restatement (what the producing rank's own writers would have written)
            nh, nw = custom_resize_shape(h, w)
            scale = (nh * 1.0 / h + nw * 1.0 / w) / 2
            g = results_json(r["general_boxes"][:n].numpy(), r["general_probs"][:n].numpy(), gc, scale, (h, w))
            s = results_json(r["specific_boxes"][:n].numpy(), r["specific_probs"][:n].numpy(), sc, scale, (h, w))
            for i, name in enumerate(names):
                if has_next or i < n - 1:
                    expect[f"flow/{seq}/{name}.flo"] = flo_bytes(r["flow"][i, :h, :w].numpy())
                both = g[i] + s[i]
                expect[f"general_proposals/{seq}/{name}.json"] = js.dumps(g[i]).encode()
                expect[f"specific_proposals/{seq}/{name}.json"] = js.dumps(s[i]).encode()
                expect[f"combined_proposals/{seq}/{name}.json"] = js.dumps(both).encode()
                ref = [dict(p) for p in both]
                for j, q in enumerate(ref):
                    q["segmentation"] = rle.encode(r["masks"][i, j, :h, :w].numpy())
                    q["conf_score"] = str(r["conf"][i, j].numpy())
                expect[f"refined_proposals/{seq}/{name}.json"] = js.dumps(ref).encode()
    assert dg.rle_overflows == ((3 if bigger else 2) if overflow else 0)      # (bigger: round 0 also decodes the other rank's chunk)
    got = {}
    for root, _, files in os.walk(out_dir):
        for f in files:
            fn = os.path.join(root, f)
            if "/wide/" not in fn:                     # (the other rank's video: not what this test restates)
                got[os.path.relpath(fn, out_dir)] = open(fn, "rb").read()
    assert sorted(got) == sorted(expect) and len(got) == 5 * T - 1
    for key in expect:
        assert got[key] == expect[key], key
    assert dg._free.qsize() == dg._n_host <= 6           # every leased host buffer came back


def test_writer_threads_run_every_call_count_their_time_and_report_the_first_error():
    from premvos_amd import io_pipeline as iop
    done, lock = [], threading.Lock()

    def work(i):
        time.sleep(0.002)
        with lock:
            done.append(i)
    wr = iop.Writer(threads=3, depth=4)
    assert wr.threads == 3
    for i in range(40):
        wr.submit(work, i)
    wr.close()
    assert sorted(done) == list(range(40)) and wr.calls == 40 and wr.busy_s >= 0.07 and 1 <= wr.max_depth <= 4
    one = iop.Writer(threads=1)
    order = []
    for i in range(20):
        one.submit(order.append, i)
    one.close()
    assert order == list(range(20))                       # one thread: submission order, as before

    def boom():
        raise ValueError("disk full")
    bad = iop.Writer(threads=2)
    bad.submit(boom)
    for i in range(5):
        bad.submit(work, i)
    with pytest.raises(ValueError):
        bad.close()
    assert iop.writer_threads() == 1 and 1 <= iop.writer_threads(merge_rank=True) <= 4


def test_merge_rank_does_not_outwait_a_failed_writer(tmp_path, monkeypatch):
    """The host buffers of the merge rank come back from the writer threads.  When a writer call fails (disk full), the calls behind
    it are skipped and their buffers never return: the exchange thread must see that and raise, not wait for a buffer for ever."""
    twins, torch = _np_twins()
    from premvos_amd import io_pipeline as iop
    from PIL import Image
    d = tmp_path / "JPEGImages" / "clip"
    d.mkdir(parents=True)
    for t in range(2):
        Image.fromarray(np.zeros((8, 10, 3), np.uint8)).save(d / f"{t:05d}.png")
    dg = stream.DeviceGather(type("P", (), {"batch": 2})(), [str(d) + "/"], [[(0, 0, 2)]], 0, 1, "cpu", **twins)
    monkeypatch.setattr(dg, "_finish_frame_py", lambda *a, **k: (_ for _ in ()).throw(OSError("No space left on device")))
    monkeypatch.setenv("PREMVOS_HOST_FILES", "py")
    r = dg.staging()
    buf = dg.x.gathered_slot(dg.x.exchange_async(r))
    wr = iop.Writer(threads=1)
    t0 = time.time()
    with pytest.raises(RuntimeError, match="file writer failed"):
        for _ in range(12):                                    # 4 buffers exist at world 1; none comes back after the failure
            dg.decode_round(0, buf, str(tmp_path / "out"), wr)
    assert time.time() - t0 < 10
    with pytest.raises(OSError):
        wr.close()
