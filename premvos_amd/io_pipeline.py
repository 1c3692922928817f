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


def creator_device():
    """Index of the calling thread's current GPU, or None on a host without one.  "The current device" is a per-thread setting and a
    new thread starts on device 0: every thread this package starts selects its CREATOR's device before it runs anything
    (``start_thread`` / ``thread_pool`` below are the only places in premvos_amd/ that create threads --
    tests/test_cpu_device_discipline.py), so "the current stream" / an index-less allocation inside a stage, lane, exchange, decode or
    writer thread of a LOCAL_RANK > 0 process means that rank's GPU, not GPU 0."""
    import torch
    return torch.cuda.current_device() if torch.cuda.is_available() else None


def bind_device(index) -> None:
    if index is not None:
        import torch
        torch.cuda.set_device(index)


def start_thread(target: Callable[[], None], name: str, daemon: bool = True) -> threading.Thread:
    """``threading.Thread(target).start()`` whose first action is to select the creating thread's GPU."""
    dev = creator_device()

    def run():
        bind_device(dev)
        target()
    t = threading.Thread(target=run, name=name, daemon=daemon)
    t.start()
    return t


def thread_pool(workers: int, prefix: str) -> ThreadPoolExecutor:
    """A ``ThreadPoolExecutor`` whose workers select the creating thread's GPU when they start."""
    return ThreadPoolExecutor(max_workers=workers, thread_name_prefix=prefix, initializer=bind_device, initargs=(creator_device(),))


def host_budget(cpus: int = None, world: int = None) -> dict:
    """Host threads of ONE rank of a one-process-per-GPU job, sized from the threads the node has per rank
    (``os.cpu_count() // LOCAL_WORLD_SIZE``; the ranks of a node share its cores).  A rank of the streaming driver runs
    main (chunking + GPU JPEG finish) + 3 stage threads + ``refine_lanes`` refinement threads + the proposal joiner + the
    writer, and ``decode`` decode-ahead threads on top (Huffman decode / PIL, GIL released).  Most of these block on the GPU or
    on files, so a rank may hold twice its share of hardware threads: with >= 6 per rank the defaults of rounds 2-3 stand (4
    decode threads, 2 refinement lanes = 12 threads per rank, 96 on an 8-GPU node of 256 threads); below that the decode pool
    shrinks first (to 1), then the second refinement lane goes.  Explicit
    PREMVOS_IO_THREADS / PREMVOS_IO_LANES / PREMVOS_STREAM_REFINE_LANES always win."""
    cpus = (os.cpu_count() or 1) if cpus is None else cpus
    if world is None:
        world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    per_rank = max(1, cpus // max(world, 1))
    budget = 2 * per_rank                         # the stage / lane threads spend most of their time blocked on the GPU or on files
    fixed = 6                                     # main, flow, proposals x2, joiner, writer
    lanes = 2 if budget >= fixed + 2 + 1 else 1
    decode = max(1, min(4, budget - fixed - lanes))
    return {"threads_per_rank": per_rank, "decode": decode, "refine_lanes": lanes, "total": fixed + lanes + decode}


def io_threads() -> int:
    v = os.environ.get("PREMVOS_IO_THREADS")
    return max(0, int(v)) if v is not None else host_budget()["decode"]


def io_lanes() -> int:
    v = os.environ.get("PREMVOS_IO_LANES")
    return max(1, int(v)) if v is not None else host_budget()["refine_lanes"]


def prefetch(jobs: Iterable[A], load: Callable[[A], B], workers: int = None, depth: int = None) -> Iterator[B]:
    """``load(job)`` for every job, in order, with up to ``depth`` loads running ahead of the consumer."""
    workers = io_threads() if workers is None else workers
    if workers <= 0:
        for j in jobs:
            yield load(j)
        return
    depth = depth or 2 * workers
    pending = collections.deque()
    with thread_pool(workers, "premvos-load") as pool:
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
    pools = [thread_pool(1, f"premvos-lane{i}") for i in range(n)]
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
    """Runs ``fn(*args)`` calls on background threads; ``close()`` waits and re-raises the first error.  One thread (the default of
    the per-stage drivers): calls run in submission order.  ``threads`` > 1: the calls share one queue and may complete in any
    order -- every call of this package writes its own file(s) and depends on no other call, so the tree is the same; the merge
    rank of a gathered multi-GPU job needs that (it writes the files of EVERY rank: ~430 frames/s x (one 3.3 MB .flo + four JSON
    files) at 8 ranks; file system calls, the RLE string packer and the JSON encoder's C core release or barely hold the
    interpreter lock).  ``busy_s`` / ``max_depth`` / ``calls``: what the threads spent inside calls and how far the queue filled."""

    runs_callables = True                        # submit(fn) with no file argument is fine: any host work may be queued here

    def __init__(self, enabled: bool = True, depth: int = 64, threads: int = None):
        self._q: "queue.Queue" = queue.Queue(maxsize=depth)
        self._err = None
        self._threads = []
        self.busy_s, self.calls, self.max_depth = 0.0, 0, 0
        self._stat_lock = threading.Lock()
        if enabled:
            n = writer_threads() if threads is None else max(1, threads)
            for i in range(n):
                self._threads.append(start_thread(self._run, f"premvos-writer{i}" if n > 1 else "premvos-writer"))

    @property
    def threads(self) -> int:
        return len(self._threads)

    def _run(self):
        import time
        while True:
            item = self._q.get()
            if item is None:
                return
            if self._err is None:
                t0 = time.perf_counter()
                try:
                    item[0](*item[1])
                except BaseException as e:          # noqa: BLE001 -- reported by close()
                    self._err = e
                dt = time.perf_counter() - t0
                with self._stat_lock:
                    self.busy_s += dt
                    self.calls += 1

    def submit(self, fn, *args):
        if not self._threads:
            fn(*args)
        else:
            self._q.put((fn, args))
            d = self._q.qsize()
            if d > self.max_depth:
                self.max_depth = d

    def close(self):
        if self._threads:
            for _ in self._threads:
                self._q.put(None)
            for t in self._threads:
                t.join()
            self._threads = []
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


def writer_threads(merge_rank: bool = False) -> int:
    """File-writer threads of one rank: PREMVOS_IO_WRITERS, else 1 -- or, for the merge rank of a gathered job (it writes every
    rank's files), up to 4 from this rank's share of the host."""
    v = os.environ.get("PREMVOS_IO_WRITERS")
    if v is not None:
        return max(1, int(v))
    return max(1, min(4, host_budget()["threads_per_rank"] // 4)) if merge_rank else 1
