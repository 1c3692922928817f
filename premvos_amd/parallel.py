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


def max_shard(n_items: int, world: int) -> int:
    return (max(n_items, 0) + world - 1) // world


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
