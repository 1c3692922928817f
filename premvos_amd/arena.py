"""Liveness-planned activation memory of one launch plan.

A plan is a fixed, stream-ordered launch list; the pointers of its tensors are frozen into the conv descriptors (and later into a
HIP graph) while the list is built.  Up to round 4 every layer owned its buffers (130 GiB for the bench's B = 16 pipeline object).
Now a plan's builder runs TWICE over the same code:

  pass 1 (``Arena(dry=True)``): ``alloc`` hands out shape-only tensors (torch "meta" device: no memory, ``data_ptr() == 0``) and
      records (bytes, first tick); ``release`` -- called by the builder once the LAST launch that reads a tensor has been appended --
      records the last tick.  The clock advances with every alloc / release, so an interval [alloc, release) is exactly the span of
      the launch list in which the tensor is live (launches execute in list order on one stream).
  ``Arena.packed()``: offsets by greedy-by-size first fit over the interval graph (two tensors may share bytes iff their intervals
      are disjoint); ONE zero-filled device buffer of the packed size.
  pass 2: the same alloc / release sequence returns views of that buffer (the sequence is checked call by call).

What stays outside the arena (own zero-filled tensor, as before): tensors whose pixel stride is wider than their channel count (the
padding channels are read by float4 loads and multiplied by zero weights: they must stay zero, never somebody else's activations),
and the resident split layout S8 of the bf16x3 mode lives in an arena of its own (bf16 pairs read as floats can be NaN).  Tensors
that are never released (a plan's named outputs: feature maps the tests and the drivers read after a run) keep their bytes to the
end of the list.  ``PREMVOS_ARENA=0``: every alloc is its own tensor again (A/B: results are bit-identical either way -- no kernel
reads bytes it or an earlier launch of the same list did not write).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch

ALIGN = 256          # bytes: every tensor starts on a 256-byte boundary (16-byte vector accesses, LDS-DMA rows)


def enabled() -> bool:
    return os.environ.get("PREMVOS_ARENA", "1") != "0"


def pack_intervals(items: List[Tuple[int, int, int]]) -> Tuple[List[int], int]:
    """items: (bytes, first tick, last tick exclusive).  Returns (offset per item, total bytes): greedy by size (largest first,
    then earliest), each block at the lowest offset where it overlaps no already placed block that is live at the same time."""
    order = sorted(range(len(items)), key=lambda i: (-items[i][0], items[i][1]))
    off = [0] * len(items)
    placed: List[int] = []
    total = 0
    for i in order:
        sz, a, b = items[i]
        sz = (sz + ALIGN - 1) // ALIGN * ALIGN
        busy = sorted((off[j], off[j] + (items[j][0] + ALIGN - 1) // ALIGN * ALIGN) for j in placed
                      if items[j][1] < b and a < items[j][2])
        pos = 0
        for lo, hi in busy:
            if pos + sz <= lo:
                break
            pos = max(pos, hi)
        off[i] = pos
        placed.append(i)
        total = max(total, pos + sz)
    return off, total


class Arena:
    def __init__(self, device, dry: bool):
        self.device, self.dry = device, dry
        self.on = enabled()
        self.tick = 0
        self.items: List[list] = []          # [bytes, first tick, last tick or None, shape, kind]
        self.cursor = 0                      # pass 2: index of the next alloc
        self.offsets: List[int] = []
        self.bufs = {}                       # kind -> backing tensor
        self.sizes = {}                      # kind -> bytes
        self.loose: List[torch.Tensor] = []  # tensors outside the arena (kept alive by the plan)
        self._ids = {}                       # id(tensor object handed out) -> item index

    # -- the builder's interface -------------------------------------------------------------------------------------
    def alloc(self, n: int, h: int, w: int, ps: int, kind: str = "f32", pooled: bool = True) -> torch.Tensor:
        """A zero-initialised (pass 1: shape-only) fp32 tensor [n,h,w,ps]; ``pooled=False``: a tensor of its own."""
        shape = (n, h, w, ps)
        if not (self.on and pooled):
            t = torch.empty(shape, dtype=torch.float32, device="meta") if self.dry else \
                torch.zeros(shape, dtype=torch.float32, device=self.device)
            self.loose.append(t)                 # (pass 1: keeps id() unique; a loose tensor may be released -- nothing to record)
            self._ids[id(t)] = -1
            return t
        nbytes = 4 * n * h * w * ps
        self.tick += 1
        if self.dry:
            self.items.append([nbytes, self.tick, None, shape, kind])
            t = torch.empty(shape, dtype=torch.float32, device="meta")
            self._ids[id(t)] = len(self.items) - 1
            self.loose.append(t)             # (keeps id() unique for the pass)
            return t
        it = self.items[self.cursor]
        assert it[3] == shape and it[4] == kind, "the plan builder's two passes diverged"
        o = self.offsets[self.cursor] // 4
        self.cursor += 1
        return self.bufs[kind][o:o + n * h * w * ps].view(n, h, w, ps)

    def side(self, shape, dtype) -> torch.Tensor:
        """A zero-filled tensor of the plan that is NOT activation memory (frames, boxes, counters, result blocks, scratch): shape-only
        in pass 1 -- the dry pass used to allocate (and drop) every one of them for real, > 1 GB for a 160-crop plan at 1080p --
        and a tensor of its own in pass 2."""
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        return torch.empty(shape, dtype=dtype, device="meta") if self.dry else torch.zeros(shape, dtype=dtype, device=self.device)

    def release(self, t) -> None:
        """The last launch reading ``t`` has been appended: its bytes may be handed to tensors allocated from now on.
        ``t`` is what ``alloc`` returned (or the ``ops.NHWC`` wrapped around exactly that tensor: full channel range, all images).
        A channel window (``NHWC.slice``) or an image range (``NHWC.images``) shares its parent's storage -- releasing one would end
        the lifetime of the whole buffer while its other readers are still in the list -- and an unknown tensor would be ignored
        silently: both are errors."""
        buf = getattr(t, "buf", t)
        if buf is not t and (t.coff != 0 or t.ps - t.c >= 4):
            raise ValueError("Arena.release: a channel window of a tensor, not the tensor alloc() returned")
        if not self.on:
            return
        self.tick += 1
        if self.dry:
            i = self._ids.get(id(buf))
            if i is None:
                raise ValueError("Arena.release: not a tensor of this arena (an image range / view, or released twice through a copy)")
            if i >= 0 and self.items[i][2] is None:
                self.items[i][2] = self.tick

    # -- between the passes ------------------------------------------------------------------------------------------
    def packed(self, shared: Optional[dict] = None) -> "Arena":
        """The real arena for pass 2.  ``shared`` (kind -> tensor, updated in place): plans that never execute concurrently and
        run stream-ordered one after the other (the refinement plans of ONE lane: one per packed slot count / frame group) live in
        the same bytes -- a plan takes the shared buffer when it is large enough, else a new one that replaces it for later plans
        (the earlier plans keep theirs).  A plan reads nothing it has not written in the same run, so whatever another plan left
        in the buffer never reaches a result."""
        assert self.dry
        real = Arena(self.device, dry=False)
        end = self.tick + 1
        real.items = [[b, a, (e if e is not None else end), s, k] for b, a, e, s, k in self.items]
        real.offsets = [0] * len(real.items)
        for kind in sorted({it[4] for it in real.items}):
            idx = [i for i, it in enumerate(real.items) if it[4] == kind]
            offs, total = pack_intervals([tuple(real.items[i][:3]) for i in idx])
            for i, o in zip(idx, offs):
                real.offsets[i] = o
            real.sizes[kind] = total
            buf = shared.get(kind) if shared is not None else None
            if buf is None or buf.numel() * 4 < total:
                buf = torch.zeros(total // 4, dtype=torch.float32, device=self.device)
                if shared is not None:
                    shared[kind] = buf
            real.bufs[kind] = buf
        return real

    def report(self) -> dict:
        """Bytes: packed arena, what one tensor per alloc would have taken, peak of simultaneously live bytes (the lower bound)."""
        naive = sum(it[0] for it in self.items)
        ev = sorted([(it[1], it[0]) for it in self.items] + [(it[2], -it[0]) for it in self.items])
        live = peak = 0
        for _, d in ev:
            live += d
            peak = max(peak, live)
        return {"arena_bytes": sum(self.sizes.values()), "one_buffer_per_tensor_bytes": naive, "peak_live_bytes": peak,
                "tensors": len(self.items), "outside_bytes": sum(4 * t.numel() for t in self.loose)}


def two_pass(device, build, shared: Optional[dict] = None):
    """``build(arena)`` twice: shapes and lifetimes, then the real thing.  Returns the real arena."""
    dry = Arena(device, dry=True)
    build(dry)
    real = dry.packed(shared)
    build(real)
    assert real.cursor == len(real.items), "the plan builder's two passes diverged"
    return real
