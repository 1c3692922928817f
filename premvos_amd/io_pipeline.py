"""Host-side overlap for the file-to-file stage drivers: once the kernels run in milliseconds the stage
time is JPEG decode, JSON parsing, RLE string packing and file writes (profiles/r01_README.md: 14 frames/s file to file vs
42 HBM-resident).  The reference does all of it serially around one ``session.run`` per box / frame
(refinement_net/forwarding/FewShotSegmentationForwarder.py:85-155, proposal_net/train.py:482-522,
optical_flow_net-PWC-Net/script_pwc_multi.py:72-103).

Three small building blocks, all order-preserving so that outputs are byte-identical to the serial drivers:

  ``prefetch(jobs, load, workers, depth)``   decode / parse ahead on a thread pool (PIL's decoder and file reads drop the GIL)
  ``lanes(items, work, n)``                  run ``work(lane, item)`` on n threads (each with its own HIP stream and its own
                                             workspace lane of the net), results in order: GPU work of one item overlaps the
                                             host post-processing (D2H sync, RLE strings) of another
  ``Writer``                                 file writes on one background thread, in submission order

PREMVOS_IO_THREADS (default 4) decode threads, PREMVOS_IO_LANES (default 2) lanes; 0 / 1 give the serial behaviour.
"""
from __future__ import annotations

import collections
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, Iterator, TypeVar

A = TypeVar("A")
B = TypeVar("B")


def io_threads() -> int:
    return max(0, int(os.environ.get("PREMVOS_IO_THREADS", "4")))


def io_lanes() -> int:
    return max(1, int(os.environ.get("PREMVOS_IO_LANES", "2")))


def prefetch(jobs: Iterable[A], load: Callable[[A], B], workers: int = None, depth: int = None) -> Iterator[B]:
    """``load(job)`` for every job, in order, with up to ``depth`` loads running ahead of the consumer."""
    workers = io_threads() if workers is None else workers
    if workers <= 0:
        for j in jobs:
            yield load(j)
        return
    depth = depth or 2 * workers
    pending = collections.deque()
    with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="premvos-load") as pool:
        for j in jobs:
            pending.append(pool.submit(load, j))
            if len(pending) >= depth:
                yield pending.popleft().result()
        while pending:
            yield pending.popleft().result()


def lanes(items: Iterable[A], work: Callable[[int, A], B], n: int = None) -> Iterator[B]:
    """``work(lane, item)`` with lane = position % n on n worker threads (one item in flight per lane); yields in order.
    Each lane always runs on the same thread, so thread-local state (the current HIP stream) stays with its lane."""
    n = io_lanes() if n is None else n
    if n <= 1:
        for it in items:
            yield work(0, it)
        return
    pools = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"premvos-lane{i}") for i in range(n)]
    pending = collections.deque()
    try:
        for k, it in enumerate(items):
            pending.append(pools[k % n].submit(work, k % n, it))
            if len(pending) >= n:
                yield pending.popleft().result()
        while pending:
            yield pending.popleft().result()
    finally:
        for p in pools:
            p.shutdown(wait=True)


class Writer:
    """Runs ``fn(*args)`` calls on one background thread in submission order; ``close()`` waits and re-raises the first error."""

    runs_callables = True                        # submit(fn) with no file argument is fine: any host work may be queued here

    def __init__(self, enabled: bool = True, depth: int = 64):
        self._q: "queue.Queue" = queue.Queue(maxsize=depth)
        self._err = None
        self._thread = None
        if enabled:
            self._thread = threading.Thread(target=self._run, name="premvos-writer", daemon=True)
            self._thread.start()

    def _run(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            if self._err is None:
                try:
                    item[0](*item[1])
                except BaseException as e:          # noqa: BLE001 -- reported by close()
                    self._err = e

    def submit(self, fn, *args):
        if self._thread is None:
            fn(*args)
        else:
            self._q.put((fn, args))

    def close(self):
        if self._thread is not None:
            self._q.put(None)
            self._thread.join()
            self._thread = None
        if self._err is not None:
            raise self._err

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        try:
            self.close()
        except BaseException:                       # noqa: BLE001
            if exc[0] is None:                      # never replace an exception that is already in flight
                raise
        return False
