"""world_size-2 gloo tests of the frame sharding + single gather (runs on CPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from premvos_amd import parallel as P


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 64, 100):
        for world in (1, 2, 3, 8):
            spans = [P.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == P.max_shard(n, world) or n == 0
    with pytest.raises(ValueError):
        P.shard_range(4, 2, 2)
    assert P.shard_videos(list("abcde"), 2, 0) == list("abc") and P.shard_videos(list("abcde"), 2, 1) == list("de")


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, e = P.shard_range(n_pairs, world, rank)
    # stand-in for the per-pair result: a [H,W,2] "flow" that encodes the pair index
    local = torch.stack([torch.full((3, 4, 2), float(t)) for t in range(s, e)]) if e > s else torch.zeros((0, 3, 4, 2))
    got = P.gather_padded(local, e - s, P.max_shard(n_pairs, world), dst=0)
    if rank == 0:
        allr = torch.cat(got)
        q.put([int(v) for v in allr[:, 0, 0, 0]])
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [5, 8, 1])
def test_gather_padded_world2_gloo(n_pairs):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == list(range(n_pairs))
