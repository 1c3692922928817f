"""Frame sharding across the GPUs of one node + the single gather to the merge rank.

The reference's only inference parallelism is a hand-edited ``curr_run_num/total_to_run`` slice of
the sorted video list, one process per slice, results meeting on the filesystem
(refinement_net/datasets/few_shot_segmentation/DAVISFewShotSegmentationDataset.py:130-150,
MergeTrack/merge.py:66-67,126-128).  Here: one process per GPU under torch.distributed (RCCL on
GPUs, gloo in the CPU tests), contiguous frame ranges per rank (pair t = frames (t, t+1), so rank r
also reads frame ``end`` as the second image of its last pair), no data-path collective, and ONE
gather of fixed-size padded buffers per chunk to the rank that runs the CPU-side merge.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, end) of ``n_items`` for ``rank``; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(max(n_items, 0), world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_videos(videos: List[str], world: int, rank: int) -> List[str]:
    """Whole videos per rank when there are at least ``world`` of them (the reference's own scheme)."""
    s, e = shard_range(len(videos), world, rank)
    return videos[s:e]


def env_rank() -> Tuple[int, int, int]:
    """(world, rank, local rank) of this process as torch.distributed.run exports them; (1, 0, 0) outside it."""
    import os
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def bind_device() -> int:
    """One process per GPU: make LOCAL_RANK's device the current one (so "cuda" means it everywhere below).  Returns its index."""
    _, _, local = env_rank()
    n = torch.cuda.device_count()
    if n:
        torch.cuda.set_device(local % n)
    return local % max(n, 1)


def my_videos(videos: List[str]) -> List[str]:
    """The stage drivers under ``python -m torch.distributed.run --nproc-per-node N -m premvos_amd.<stage>.driver ...``: rank r
    takes the r-th contiguous slice of the video list -- the reference's ``curr_run_num / total_to_run`` scheme
    (DAVISFewShotSegmentationDataset.py:130-150, merge.py:66-67,126-128) with the slice picked by the launcher instead of by
    editing the file; results meet on the filesystem, no process group is needed.  Outside torch.distributed.run: all videos."""
    world, rank, _ = env_rank()
    return shard_videos(list(videos), world, rank) if world > 1 else list(videos)


def max_shard(n_items: int, world: int) -> int:
    return (max(n_items, 0) + world - 1) // world


def balance_videos(frame_counts: List[int], world: int, capacity: Optional[List[float]] = None) -> List[List[int]]:
    """Whole videos per rank, longest video first onto the least loaded rank (ties: lower rank, lower index) -- the same
    granularity as the reference's hand-edited list slices (``shard_videos``), but balanced by frame count.  ``capacity``: relative
    speed of every rank (default all 1): a rank's load counts as frames / capacity, a rank of capacity 0 gets nothing.
    Deterministic; returns the video indices of every rank in ascending order."""
    cap = [1.0] * world if capacity is None else list(capacity)
    live = [r for r in range(world) if cap[r] > 0]
    if not live:
        raise ValueError("no rank with a positive capacity")
    load, out = [0] * world, [[] for _ in range(world)]
    for v in sorted(range(len(frame_counts)), key=lambda i: (-frame_counts[i], i)):
        r = min(live, key=lambda k: ((load[k] + frame_counts[v]) / cap[k], k)) if capacity is not None else \
            min(range(world), key=lambda k: (load[k], k))
        load[r] += frame_counts[v]
        out[r].append(v)
    return [sorted(x) for x in out]


def weighted_ranges(n_items: int, capacity: List[float]) -> List[Tuple[int, int]]:
    """[start, end) of ``n_items`` for every rank, sizes proportional to ``capacity`` (largest-remainder rounding, ties to the lower
    rank): the contiguous split of ``shard_range`` for ranks of unequal speed."""
    total = float(sum(capacity))
    if total <= 0:
        raise ValueError("no rank with a positive capacity")
    exact = [max(n_items, 0) * c / total for c in capacity]
    size = [int(e) for e in exact]
    for r in sorted(range(len(capacity)), key=lambda k: (-(exact[k] - size[k]), k))[:max(n_items, 0) - sum(size)]:
        size[r] += 1
    out, start = [], 0
    for sz in size:
        out.append((start, start + sz))
        start += sz
    return out


def plan_shards(frame_counts: List[int], world: int, rank: int, chunk: int,
                scheme: str = "balanced", merge_share: float = 1.0, merge_rank: int = 0) -> List[Tuple[int, int, int]]:
    """The work of ``rank``: a list of (video index, first frame, end frame).

    * at least ``world`` videos: whole videos per rank (``scheme`` = "balanced": by frame count; "contiguous": the reference's
      own slice of the sorted list, DAVISFewShotSegmentationDataset.py:130-150 / merge.py:66-67,126-128);
    * fewer videos than ranks: every video is cut into contiguous frame ranges.  The cuts fall on multiples of ``chunk`` (the
      driver's launch batch), i.e. ranks share out the CHUNKS a one-rank run would form: every chunk then holds the same
      frames -- hence runs the same kernels on the same data and writes the same bytes -- whatever the number of ranks.
      Frame pair t = (t, t+1) belongs to the rank that owns frame t, so a rank whose range ends before the video does also
      READS frame ``end`` (second image of its last pair, script_pwc_multi.py:100-102); it writes nothing for it.
      Successive videos start their split at a rotated rank, so the ranks holding one chunk more are not always the first ones.
    * ``merge_share`` < 1 (round 6): ``merge_rank`` -- the rank that also turns every rank's gathered results into files
      (stream.DeviceGather) -- is planned as a rank of that relative speed: it computes its own chunks ~4 % slower beside the other
      ranks' copies and writer traffic (profiles/r06_merge_ingest.json), and a statically balanced job is as fast as its slowest rank.
      0 = the merge rank computes nothing.  Which frames a rank owns never changes what is computed for a frame.
    """
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    if chunk <= 0:
        raise ValueError("chunk must be positive")
    if not (0.0 <= merge_share <= 1.0) or not (0 <= merge_rank < world):
        raise ValueError("merge_share must lie in [0, 1], merge_rank in [0, world)")
    if world == 1:
        merge_share = 1.0
    capacity = None if merge_share == 1.0 else [merge_share if r == merge_rank else 1.0 for r in range(world)]
    nv = len(frame_counts)
    live = world if capacity is None or merge_share > 0 else world - 1
    if nv >= live:
        if scheme == "contiguous":
            s, e = shard_range(nv, world, rank) if capacity is None else weighted_ranges(nv, capacity)[rank]
            mine = list(range(s, e))
        elif scheme == "balanced":
            mine = balance_videos(frame_counts, world, capacity)[rank]
        else:
            raise ValueError(f"unknown sharding scheme {scheme!r}")
        return [(v, 0, frame_counts[v]) for v in mine if frame_counts[v] > 0]
    out, rot = [], 0
    for v, n in enumerate(frame_counts):
        n_chunks = -(-n // chunk)
        if capacity is None:
            cs, ce = shard_range(n_chunks, world, (rank + rot) % world)
            rot = (rot + world - n_chunks % world) % world
        else:                      # (no rotation: the weighted split already decides who holds the odd chunk)
            cs, ce = weighted_ranges(n_chunks, capacity)[rank]
        if ce > cs:
            out.append((v, cs * chunk, min(ce * chunk, n)))
    return out


def gather_padded(local: torch.Tensor, n_valid: int, capacity: int, dst: int = 0,
                  group=None) -> Optional[List[torch.Tensor]]:
    """Gather per-rank results of differing length: every rank contributes ``capacity`` rows
    (zero padded) plus its valid count; ``dst`` gets the per-rank tensors trimmed to their counts."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [local[:n_valid]]
    rank = dist.get_rank(group)
    pad = torch.zeros((capacity,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:n_valid] = local[:n_valid]
    cnt = torch.tensor([n_valid], dtype=torch.int64, device=local.device)
    bufs = cnts = None
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        cnts = [torch.empty_like(cnt) for _ in range(world)]
    dist.gather(pad, bufs, dst=dst, group=group)
    dist.gather(cnt, cnts, dst=dst, group=group)
    if rank != dst:
        return None
    return [b[:int(c.item())] for b, c in zip(bufs, cnts)]


class ResultExchange:
    """The one exchange of the path: every rank's per-step results go to the merge rank as ONE fixed-size byte buffer in
    ONE gather.  Layout (round 6: what the merge rank's HOST needs comes first, in one contiguous prefix):

        flow [B,H,W,2] f32 | boxes / scores / conf / counts f32 | RLE offsets int32 [B*P + 1] | RLE pool int32 | masks [B,P,H,W] bit-packed

    The run boundaries of every mask (column-major, the order COCO RLE counts them in) are found on the PRODUCING rank's GPU while
    it packs (premvos_rle_boundaries_pooled_u8: one variable-length pool per chunk, ``rle_runs`` entries per mask slot on average),
    so the merge rank turns a gathered buffer into files with one device-to-host copy of the prefix and host work only -- no
    unpack, no kernel, no per-frame synchronisation on ITS GPU (round 5 did all of that there, for every rank's chunk; the reference
    encodes on the CPU of whoever ran the net: FewShotSegmentationForwarder.py:137-143).  The bit-packed masks still travel: they
    are what an in-process merge consumes (mergetrack.py) and the fallback when a chunk's boundaries overflow the pool.

    ``pack_bits`` / ``unpack_bits`` / ``rle_pool`` default to the HIP kernels; the CPU gloo tests inject numpy twins.  With the gloo
    backend (CPU tests; several ranks sharing one GPU when exercising the logic by hand) the buffer is staged through the host.
    """

    SMALL_COLS_FIXED = 2 * 20 * 5 + 2          # general + specific: 20 boxes x (4 + prob), + the two detection counts

    def __init__(self, batch: int, h: int, w: int, boxes_per_frame: int, device, dst: int = 0, group=None,
                 pack_bits=None, unpack_bits=None, slots: int = 2, rle_pool=None, rle_runs: Optional[int] = None):
        import os
        self.B, self.H, self.W, self.P, self.dst, self.group = batch, h, w, boxes_per_frame, dst, group
        self.device = torch.device(device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.flow_bytes = batch * h * w * 2 * 4
        self.mask_bits = batch * boxes_per_frame * h * w
        self.mask_bytes = (self.mask_bits + 7) // 8
        self.small_cols = self.SMALL_COLS_FIXED + boxes_per_frame
        self.n_masks = batch * boxes_per_frame
        # boundaries per mask slot, on average over the chunk (an object mask of a 480p frame has a few hundred to ~2000; empty
        # slots have none); a chunk that needs more sets offsets[-1] > pool_cap and is encoded from its masks on the merge rank
        runs = int(os.environ.get("PREMVOS_GATHER_RLE_RUNS", "2048")) if rle_runs is None else rle_runs
        self.pool_cap = max(16, self.n_masks * runs)
        self.off_small = (self.flow_bytes + 15) // 16 * 16
        self.off_rle_off = self.off_small + batch * self.small_cols * 4
        self.off_rle_pool = (self.off_rle_off + (self.n_masks + 1) * 4 + 15) // 16 * 16
        self.off_mask = (self.off_rle_pool + self.pool_cap * 4 + 15) // 16 * 16
        self.prefix_bytes = self.off_mask           # everything the merge rank's host reads
        self.nbytes = self.off_mask + (self.mask_bytes + 15) // 16 * 16
        # two slots: the gather of step i (async) is still reading slot i % 2 while step i + 1 is packed into the other one
        self._packed = [torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device) for _ in range(slots)]
        self.packed = self._packed[0]
        self.host_staged = dist.is_initialized() and dist.get_backend(group) == "gloo" and self.device.type != "cpu"
        gdev = torch.device("cpu") if self.host_staged else self.device
        self._gathered = [([torch.zeros(self.nbytes, dtype=torch.uint8, device=gdev) for _ in range(self.world)]
                           if self.rank == dst else None) for _ in range(slots)]
        self.gathered = self._gathered[0]
        self._work = [None] * slots
        self._send = [None] * slots
        self._n = 0
        self.wait_s = 0.0          # seconds this rank spent blocked in wait() / flush() (bench.py: per-rank scaling diagnostics)
        self._pack_bits, self._unpack_bits = pack_bits or _hip_pack_bits, unpack_bits or _hip_unpack_bits
        self._rle_pool = rle_pool or _hip_rle_pool
        self._rle_ws = None

    def _views(self, buf: torch.Tensor):
        B = self.B
        return (buf[:self.flow_bytes].view(torch.float32).view(B, self.H, self.W, 2),
                buf[self.off_small:self.off_rle_off].view(torch.float32).view(B, self.small_cols),
                buf[self.off_rle_off:self.off_rle_off + (self.n_masks + 1) * 4].view(torch.int32),
                buf[self.off_rle_pool:self.off_rle_pool + self.pool_cap * 4].view(torch.int32),
                buf[self.off_mask:self.off_mask + self.mask_bytes])

    def pack(self, r, slot: int = 0) -> torch.Tensor:
        """r: the dict FramePipeline.step returns (``"hw"``: the frames' own size when it is smaller than the buffer's H x W --
        the RLE runs over that window).  Fills and returns this rank's byte buffer."""
        B = self.B
        self.packed = self._packed[slot]
        flow, small, roff, rpool, bits = self._views(self.packed)
        flow.copy_(r["flow"])
        masks = r["masks"].contiguous()
        h, w = r.get("hw", (self.H, self.W))
        self._rle_pool(masks.view(self.n_masks, self.H, self.W), int(h), int(w), rpool, roff, self)
        self._pack_bits(masks.view(-1), bits)
        c = 0
        for key, n in (("general_boxes", 80), ("general_probs", 20), ("specific_boxes", 80), ("specific_probs", 20),
                       ("conf", self.P), ("general_count", 1), ("specific_count", 1)):
            small[:, c:c + n].copy_(r[key].reshape(B, n))        # counts are stored as exact small floats
            c += n
        return self.packed

    def exchange(self, r):
        """pack + the single gather.  Returns the list of per-rank byte buffers on the merge rank, None elsewhere."""
        buf = self.pack(r)
        if self.world == 1 and not dist.is_initialized():
            return [buf]
        send = buf.cpu() if self.host_staged else buf
        dist.gather(send, self.gathered if self.rank == self.dst else None, dst=self.dst, group=self.group)
        return self.gathered

    def exchange_async(self, r):
        """pack + the single gather, NOT waited for: the collective of step i runs while step i + 1 computes (RCCL: on its own
        stream behind the pack kernels; gloo: on the backend's thread).  A slot is reused every ``slots`` calls, after waiting for
        the gather that last used it.  Returns the slot; ``wait(slot)`` (or ``flush()``) makes ``gathered_slot(slot)`` valid."""
        slot = self._n % len(self._packed)
        self._n += 1
        self.wait(slot)
        buf = self.pack(r, slot)
        if not dist.is_initialized():
            self._gathered[slot] = [buf]
            return slot
        send = self._send[slot] = (buf.cpu() if self.host_staged else buf)      # kept alive until the gather is done
        self._work[slot] = dist.gather(send, self._gathered[slot] if self.rank == self.dst else None, dst=self.dst,
                                       group=self.group, async_op=True)
        return slot

    def wait(self, slot: int):
        w, self._work[slot] = self._work[slot], None
        if w is not None:
            import time
            t = time.perf_counter()
            w.wait()
            self.wait_s += time.perf_counter() - t

    def flush(self):
        for s in range(len(self._work)):
            self.wait(s)

    def gathered_slot(self, slot: int):
        return self._gathered[slot]

    def unpack(self, buf: torch.Tensor, masks: bool = True):
        """One rank's byte buffer -> dict of tensors on ``buf``'s device (``buf`` may be just the first ``prefix_bytes`` bytes when
        ``masks`` is False: that is what the merge rank copies to its host).  ``masks``: also unpack the bit-packed masks to
        {0,1} bytes (a kernel on the GPU).  ``rle_offsets`` [B*P + 1] / ``rle_pool``: mask slot i's ascending column-major run
        boundaries are rle_pool[rle_offsets[i] : rle_offsets[i + 1]] -- valid unless rle_offsets[-1] > pool_cap."""
        B, P = self.B, self.P
        flow = buf[:self.flow_bytes].view(torch.float32).view(B, self.H, self.W, 2)
        small = buf[self.off_small:self.off_rle_off].view(torch.float32).view(B, self.small_cols)
        out = {"flow": flow,
               "rle_offsets": buf[self.off_rle_off:self.off_rle_off + (self.n_masks + 1) * 4].view(torch.int32),
               "rle_pool": buf[self.off_rle_pool:self.off_rle_pool + self.pool_cap * 4].view(torch.int32)}
        if masks:
            m = torch.empty(self.mask_bits, dtype=torch.uint8, device=buf.device)
            self._unpack_bits(buf[self.off_mask:self.off_mask + self.mask_bytes], m)
            out["masks"] = m.view(B, P, self.H, self.W)
        c = 0
        for key, n, shape in (("general_boxes", 80, (B, 20, 4)), ("general_probs", 20, (B, 20)), ("specific_boxes", 80, (B, 20, 4)),
                              ("specific_probs", 20, (B, 20)), ("conf", P, (B, P)), ("general_count", 1, (B,)),
                              ("specific_count", 1, (B,))):
            out[key] = small[:, c:c + n].reshape(shape)
            c += n
        out["general_count"], out["specific_count"] = out["general_count"].to(torch.int32), out["specific_count"].to(torch.int32)
        return out


def _hip_rle_pool(masks: torch.Tensor, h: int, w: int, pool: torch.Tensor, offsets: torch.Tensor, x: "ResultExchange"):
    """masks uint8 [n, H, W] (CUDA): the run boundaries of every mask's h x w window -> pool / offsets (views of the packed buffer)."""
    from . import _lib
    _lib.require_gpu()
    assert masks.is_cuda and pool.is_cuda and offsets.is_cuda and masks.dtype == torch.uint8
    n, H, W = masks.shape
    lib = _lib.load()
    need = (int(lib.premvos_rle_workspace_bytes(n, h, w)) + 3) // 4
    if x._rle_ws is None or x._rle_ws.numel() < need or x._rle_ws.device != masks.device:
        x._rle_ws = torch.empty(need, dtype=torch.int32, device=masks.device)
    _lib.check(lib.premvos_rle_boundaries_pooled_u8(masks.data_ptr(), n, h, w, H * W, W, pool.data_ptr(), pool.numel(),
                                                    offsets.data_ptr(), x._rle_ws.data_ptr(), _lib.current_stream()), "rle_boundaries_pooled")


def _hip_pack_bits(masks: torch.Tensor, out: torch.Tensor):
    from . import _lib
    _lib.require_gpu()
    assert masks.is_cuda and out.is_cuda and masks.dtype == torch.uint8 and out.numel() == (masks.numel() + 7) // 8
    _lib.check(_lib.load().premvos_mask_pack_bits_u8(masks.data_ptr(), masks.numel(), out.data_ptr(), _lib.current_stream()),
               "mask_pack_bits")


def _hip_unpack_bits(bits: torch.Tensor, out: torch.Tensor):
    from . import _lib
    _lib.require_gpu()
    if not bits.is_cuda:                       # host-staged (gloo) buffers are unpacked where the merge loop's GPU helpers live
        bits = bits.cuda()
        tmp = torch.empty(out.numel(), dtype=torch.uint8, device=bits.device)
        _lib.check(_lib.load().premvos_mask_unpack_bits_u8(bits.data_ptr(), out.numel(), tmp.data_ptr(), _lib.current_stream()),
                   "mask_unpack_bits")
        out.copy_(tmp)
        return
    _lib.check(_lib.load().premvos_mask_unpack_bits_u8(bits.data_ptr(), out.numel(), out.data_ptr(), _lib.current_stream()),
               "mask_unpack_bits")
