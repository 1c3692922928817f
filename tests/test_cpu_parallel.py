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


# ---------------------------------------------------------------------------------------------------------------------
# bench.py's own exchange (premvos_amd.parallel.ResultExchange: ONE gather of one packed buffer per rank), world size 2
def _np_pack_bits(masks, out):
    import numpy as np
    out.copy_(torch.from_numpy(np.packbits((masks.numpy() != 0), bitorder="little")))


def _np_unpack_bits(bits, out):
    import numpy as np
    out.copy_(torch.from_numpy(np.unpackbits(bits.numpy(), bitorder="little")[:out.numel()]))


def _fake_results(rank, B, P_, H, W):
    g = torch.Generator().manual_seed(100 + rank)
    return {"flow": torch.randn((B, H, W, 2), generator=g), "masks": (torch.rand((B, P_, H, W), generator=g) > 0.6).to(torch.uint8),
            "conf": torch.rand((B, P_), generator=g), "general_boxes": torch.rand((B, 20, 4), generator=g) * 100,
            "general_probs": torch.rand((B, 20), generator=g), "general_count": torch.tensor([rank + 3] * B, dtype=torch.int32),
            "specific_boxes": torch.rand((B, 20, 4), generator=g) * 50, "specific_probs": torch.rand((B, 20), generator=g),
            "specific_count": torch.tensor([20 - rank] * B, dtype=torch.int32)}


def _xchg_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, P_, H, W = 2, 3, 5, 7               # 2*3*5*7 = 210 mask bits: not a multiple of 8
    x = P.ResultExchange(B, H, W, P_, "cpu", pack_bits=_np_pack_bits, unpack_bits=_np_unpack_bits)
    calls = {"n": 0}
    real_gather = dist.gather

    def counting_gather(*a, **k):
        calls["n"] += 1
        return real_gather(*a, **k)
    dist.gather = counting_gather
    got = None
    for step in range(2):                   # buffers are reused step after step
        got = x.exchange(_fake_results(rank, B, P_, H, W))
    assert calls["n"] == 2                  # exactly ONE gather per step
    if rank == 0:
        ok = len(got) == world
        for r in range(world):
            ref, u = _fake_results(r, B, P_, H, W), x.unpack(got[r])
            ok = ok and all(torch.equal(u[k], ref[k]) for k in ref)
        q.put(ok)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_result_exchange_single_packed_gather_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_xchg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    assert q.get(timeout=120) is True
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0


def test_result_exchange_layout_is_fixed_size_and_bit_packed():
    x = P.ResultExchange(16, 480, 854, 20, "cpu", pack_bits=_np_pack_bits, unpack_bits=_np_unpack_bits)
    assert x.flow_bytes == 16 * 480 * 854 * 2 * 4 and x.mask_bytes == 16 * 20 * 480 * 854 // 8
    assert x.off_small % 16 == 0 and x.nbytes == x.off_small + 16 * (2 * 20 * 5 + 2 + 20) * 4
    assert x.nbytes < 0.16 * (x.flow_bytes + 16 * 20 * 480 * 854 + 16 * 222 * 4) + x.flow_bytes      # masks shrink 8x
