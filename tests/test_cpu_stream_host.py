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
