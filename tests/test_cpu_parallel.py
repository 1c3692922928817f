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


def _np_rle_pool(masks, h, w, pool, offsets, x):
    """numpy twin of premvos_rle_boundaries_pooled_u8: ascending column-major change positions of every mask's h x w window."""
    import numpy as np
    m = masks.numpy()
    run, off, cap = 0, [0], pool.numel()
    for i in range(m.shape[0]):
        flat = (m[i, :h, :w] != 0).reshape(-1, order="F").astype(np.int8)
        pos = np.flatnonzero(np.diff(np.concatenate(([0], flat)))).astype(np.int32)
        keep = pos[:max(0, min(len(pos), cap - run))]
        pool[run:run + len(keep)] = torch.from_numpy(keep)
        run += len(pos)
        off.append(run)
    offsets.copy_(torch.tensor(off, dtype=torch.int32))


def _xchg(B, H, W, P_, **kw):
    return P.ResultExchange(B, H, W, P_, "cpu", pack_bits=_np_pack_bits, unpack_bits=_np_unpack_bits, rle_pool=_np_rle_pool, **kw)


def _check_rle(u, ref, x):
    """The run boundaries a buffer carries decode to the masks it carries (premvos_amd.rle's own reader as the witness)."""
    import numpy as np
    from premvos_amd import rle
    off, pool = u["rle_offsets"].numpy(), u["rle_pool"].numpy()
    assert off[0] == 0 and off[-1] <= x.pool_cap
    masks = ref["masks"].numpy().reshape(x.n_masks, x.H, x.W)
    for i in range(x.n_masks):
        edges = np.concatenate(([0], pool[off[i]:off[i + 1]].astype(np.int64), [x.H * x.W]))
        back = rle.decode({"size": [x.H, x.W], "counts": [int(c) for c in np.diff(edges)]})
        if not np.array_equal(back, masks[i]):
            return False
    return True


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
    x = _xchg(B, H, W, P_)
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
            ok = ok and all(torch.equal(u[k], ref[k]) for k in ref) and _check_rle(u, ref, x)
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
    x = _xchg(16, 480, 854, 20)
    assert x.flow_bytes == 16 * 480 * 854 * 2 * 4 and x.mask_bytes == 16 * 20 * 480 * 854 // 8
    # what the merge rank's HOST reads is one contiguous prefix -- flow, detections / conf / counts, run-boundary offsets + pool --
    # and the bit-packed masks come last
    assert x.off_small % 16 == 0 and x.off_rle_off == x.off_small + 16 * (2 * 20 * 5 + 2 + 20) * 4
    assert x.off_rle_pool % 16 == 0 and x.off_rle_pool >= x.off_rle_off + (16 * 20 + 1) * 4
    assert x.off_mask % 16 == 0 and x.off_mask == x.prefix_bytes >= x.off_rle_pool + x.pool_cap * 4
    assert x.pool_cap == 16 * 20 * 2048 and x.nbytes == x.off_mask + x.mask_bytes
    assert x.nbytes < 0.16 * (x.flow_bytes + 16 * 20 * 480 * 854 + 16 * 222 * 4) + x.flow_bytes + 4 * x.pool_cap + 4096   # masks shrink 8x


def test_result_exchange_run_boundaries_window_overflow_and_strings():
    """The pooled run boundaries a buffer carries: over the frames' own h x w window of a larger block, flagged (offsets[-1] >
    pool_cap) when the pool is too small, and turned into COCO strings by the C packer exactly as rle.encode does from the mask."""
    import numpy as np
    from premvos_amd import rle
    B, P_, H, W, h, w = 2, 3, 9, 11, 7, 8
    r = _fake_results(5, B, P_, H, W)
    r["masks"][0, 0] = 0                                     # an empty slot: one run
    r["masks"][0, 1] = 1                                     # all foreground: the first run has length 0
    r["hw"] = (h, w)
    x = _xchg(B, H, W, P_, rle_runs=64)
    u = x.unpack(x.exchange(r)[0])
    off, pool = u["rle_offsets"].numpy(), u["rle_pool"].numpy()
    assert off[-1] <= x.pool_cap
    strings = rle.strings_from_pool(pool, off, h * w)
    m = r["masks"].numpy().reshape(B * P_, H, W)[:, :h, :w]
    for i in range(B * P_):
        assert strings[i] == rle.encode(m[i])["counts"], i
        assert np.array_equal(rle.decode({"size": [h, w], "counts": strings[i]}), m[i])
    # a window of the strings (the slots of one frame), as the merge rank asks for them
    assert rle.strings_from_pool(pool, off[P_:2 * P_ + 1], h * w) == strings[P_:]
    assert rle.strings_from_pool(pool, off[:1], h * w) == []
    # too small a pool: the total still says how many there were
    y = _xchg(B, H, W, P_, rle_runs=1)
    v = y.unpack(y.exchange(r)[0])
    assert int(v["rle_offsets"][-1]) == int(off[-1]) > y.pool_cap


# ---------------------------------------------------------------------------------------------------------------------
# the product's sharding (premvos_amd.stream.run -> parallel.plan_shards + stream.iter_chunks)
def test_plan_shards_with_a_slower_or_idle_merge_rank():
    """``merge_share`` (round 6): the rank that also writes every rank's gathered files is planned as a slower rank.  Whatever the
    share, every frame is owned exactly once, cuts stay on chunk boundaries (same chunks as the one-rank run), and share 1 is the
    old plan."""
    import random
    rng = random.Random(3)
    for _ in range(60):
        world, chunk = rng.randint(2, 9), rng.choice([2, 8, 16])
        counts = [rng.randint(0, 200) for _ in range(rng.randint(1, 14))]
        for share in (1.0, 0.9, 0.5, 0.0):
            for scheme in ("balanced", "contiguous"):
                plans = [P.plan_shards(counts, world, r, chunk, scheme, merge_share=share) for r in range(world)]
                if share == 1.0:
                    assert plans == [P.plan_shards(counts, world, r, chunk, scheme) for r in range(world)]
                owned = {}
                for r, pl in enumerate(plans):
                    for v, a, b in pl:
                        assert 0 <= a < b <= counts[v] and a % chunk == 0 and (b % chunk == 0 or b == counts[v])
                        for t in range(a, b):
                            assert (v, t) not in owned
                            owned[(v, t)] = r
                assert len(owned) == sum(counts)
                if share == 0.0:
                    assert plans[0] == []
    # one long video over 8 ranks: 125 chunks, rank 0 planned at 0.9 -> 14 chunks where the others hold 15 or 16
    plans = [P.plan_shards([1000], 8, r, 8, merge_share=0.9) for r in range(8)]
    sizes = [-(-(b - a) // 8) for pl in plans for _, a, b in pl]
    assert sizes[0] == 14 and sorted(sizes[1:]) == [15, 16, 16, 16, 16, 16, 16] and sum(sizes) == 125
    # whole videos: the merge rank's frames / share is what gets balanced
    plans = [P.plan_shards([100] * 15 + [50], 4, r, 8, merge_share=0.5) for r in range(4)]
    frames = [sum(b - a for _, a, b in pl) for pl in plans]
    assert frames[0] <= 0.6 * min(frames[1:]) and sum(frames) == 1550
    assert P.weighted_ranges(10, [0.9, 1, 1, 1]) == [(0, 2), (2, 5), (5, 8), (8, 10)] and P.weighted_ranges(0, [1, 1]) == [(0, 0), (0, 0)]
    with pytest.raises(ValueError):
        P.plan_shards([10], 2, 0, 8, merge_share=1.5)


def test_plan_shards_whole_videos_and_chunk_aligned_ranges():
    # at least as many videos as ranks: whole videos, balanced by frame count (or the reference's contiguous slices)
    counts = [70, 50, 30, 90, 20]
    plans = [P.plan_shards(counts, 3, r, 8) for r in range(3)]
    assert sorted(x for p in plans for x in p) == [(v, 0, n) for v, n in enumerate(counts)]
    assert max(sum(e - s for _, s, e in p) for p in plans) == 90            # 90 | 70+20 | 50+30
    assert [P.plan_shards(counts, 2, r, 8, "contiguous") for r in range(2)] == [[(0, 0, 70), (1, 0, 50), (2, 0, 30)],
                                                                              [(3, 0, 90), (4, 0, 20)]]
    # fewer videos than ranks: every video is cut at multiples of the chunk; the ranges tile [0, n) without gaps or overlaps
    for world in (2, 3, 8, 11):
        for counts in ([70, 50], [7], [1], [16, 0, 3]):
            plans = [P.plan_shards(counts, world, r, 8) for r in range(world)] if len(counts) < world else None
            if plans is None:
                continue
            for v, n in enumerate(counts):
                rs = sorted((s, e) for p in plans for vv, s, e in p if vv == v)
                assert all(s % 8 == 0 for s, _ in rs) and all(e % 8 == 0 or e == n for _, e in rs)
                assert [s for s, _ in rs] == [0] * bool(rs) + [e for _, e in rs][:-1] and (not rs or rs[-1][1] == n)
                assert bool(rs) == (n > 0)
    # the extra chunk does not always land on the first ranks: 2 videos of 9 chunks over 8 ranks
    plans = [P.plan_shards([72, 72], 8, r, 8) for r in range(8)]
    assert sorted(sum(e - s for _, s, e in p) for p in plans) == [16] * 6 + [24] * 2
    with pytest.raises(ValueError):
        P.plan_shards([5], 2, 2, 8)


def _shard_worker(rank, world, port, q, n_frames, batch):
    """Every rank walks ITS shards with the product's chunk iterator and a fake decoder (frame = its index); the 'stages' are
    what the real ones do with a chunk: one proposal result per owned frame, one flow pair per (frame, successor)."""
    import numpy as np
    from premvos_amd import stream
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    images = [f"/video/{t:05d}.jpg" for t in range(n_frames)]
    decoded = []

    def load(fn):
        decoded.append(fn)
        return np.full((2, 2, 3), int(os.path.basename(fn)[:5]), dtype=np.int32)
    owned, pairs, chunk_sizes = [], [], []
    for _, first, end in P.plan_shards([n_frames], world, rank, batch):
        for names, frames, nxt in stream.iter_chunks(images, first, end, batch, load):
            ids = [int(f[0, 0, 0]) for f in frames]
            assert [int(n) for n in names] == ids
            owned += ids
            chunk_sizes.append(len(ids))
            second = ids[1:] + ([int(nxt[0, 0, 0])] if nxt is not None else [])
            pairs += list(zip(ids, second))
    assert len(decoded) == len(set(decoded))                                   # every frame decoded once per rank
    res = torch.tensor([[a, b] for a, b in pairs] or [[-1, -1]], dtype=torch.int64)
    got = P.gather_padded(res, len(pairs), P.max_shard(-(-n_frames // batch), world) * batch)
    own = [None] * world
    dist.all_gather_object(own, (owned, chunk_sizes, len(decoded)))
    if rank == 0:
        q.put(([g.tolist() for g in got], own))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,batch,world", [(7, 2, 2), (16, 4, 2), (5, 8, 2)])
def test_frame_range_sharding_with_boundary_frames_world2_gloo(n_frames, batch, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q, n_frames, batch)) for r in range(world)]
    for p in procs:
        p.start()
    pairs, own = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # every frame is owned by exactly one rank; every pair (t, t+1) is computed exactly once, the boundary ones included
    assert sorted(t for o, _, _ in own for t in o) == list(range(n_frames))
    assert sorted(tuple(x) for g in pairs for x in g) == [(t, t + 1) for t in range(n_frames - 1)]
    # the chunks are the ones a single rank forms (cuts at multiples of the batch): same launch shapes whatever the world size
    assert sorted(c for _, cs, _ in own for c in cs) == sorted([batch] * (n_frames // batch) + [n_frames % batch] * bool(n_frames % batch))
    # a rank whose range stops inside the video decodes exactly one frame more than it owns
    for r, (o, _, ndec) in enumerate(own):
        assert ndec == len(o) + (1 if o and o[-1] != n_frames - 1 else 0)


def _async_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, P_, H, W = 2, 3, 5, 7
    x = _xchg(B, H, W, P_)
    ok, slots = True, []
    for step in range(5):                   # double-buffered: the gather of step i is waited for when slot i % 2 comes round again
        slots.append(x.exchange_async(_fake_results(10 * step + rank, B, P_, H, W)))
        if step >= 1 and rank == 0:         # "one step later": the previous step's buffers are complete and untouched by this step
            x.wait(slots[step - 1])
            for r in range(world):
                ref, u = _fake_results(10 * (step - 1) + r, B, P_, H, W), x.unpack(x.gathered_slot(slots[step - 1])[r])
                ok = ok and all(torch.equal(u[k], ref[k]) for k in ref)
    x.flush()
    if rank == 0:
        q.put(ok and slots == [0, 1, 0, 1, 0])
    dist.barrier()
    dist.destroy_process_group()


def test_result_exchange_async_double_buffered_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_async_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    assert q.get(timeout=120) is True
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0


def test_stage_drivers_take_the_reference_slice_of_the_video_list(monkeypatch):
    """`torch.distributed.run -m premvos_amd.<stage>.driver`: rank r works on the r-th contiguous slice of the video list
    (curr_run_num / total_to_run of DAVISFewShotSegmentationDataset.py:130-150), all of it outside the launcher."""
    vids = [f"v{i}" for i in range(7)]
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert P.my_videos(vids) == vids and P.env_rank() == (1, 0, 0)
    got = []
    for r in range(3):
        monkeypatch.setenv("WORLD_SIZE", "3")
        monkeypatch.setenv("RANK", str(r))
        monkeypatch.setenv("LOCAL_RANK", str(r))
        got.append(P.my_videos(vids))
    assert got == [vids[0:3], vids[3:5], vids[5:7]] and P.env_rank() == (3, 2, 2)


# ---------------------------------------------------------------------------------------------------------------------
# world 8 (one node of MI355X): the same paths with ranks that own nothing
def _spawn(target, world, *args, timeout=240):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=timeout)
    for p in procs:
        p.join(timeout=timeout)
        assert p.exitcode == 0
    return res


def _strong_worker(rank, world, port, q, T, B):
    """bench.py's strong-scaling step at world 8: ONE video of T frame pairs in chunks of B shared out by plan_shards; every rank
    takes part in every round's gather, the ranks that own no chunk of the round send whatever they packed last (bench.py:381-387);
    the merge rank takes a chunk from the rank the plan names."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P_, H, W = 3, 5, 7
    x = _xchg(B, H, W, P_)
    plans = [P.plan_shards([T], world, r, B) for r in range(world)]
    chunks = [[c0 // B for _, s0, e0 in p for c0 in range(s0, e0, B)] for p in plans]      # chunk ids per rank
    rounds = max(len(c) for c in chunks)
    ok = True
    for step in range(2):                                  # two passes over the video: the slots come round again
        last, slots = _fake_results(999, B, P_, H, W), []  # an idle rank's filler
        for k in range(rounds):
            if k < len(chunks[rank]):
                last = _fake_results(1000 * step + chunks[rank][k], B, P_, H, W)
            slots.append(x.exchange_async(last))
        x.flush()
        if rank == 0:                                      # the last `len(x._packed)` rounds are still in their slots
            for k in range(max(0, rounds - 2), rounds):
                for r in range(world):
                    if k < len(chunks[r]):
                        ref, u = _fake_results(1000 * step + chunks[r][k], B, P_, H, W), x.unpack(x.gathered_slot(slots[k])[r])
                        ok = ok and all(torch.equal(u[key], ref[key]) for key in ref)
    if rank == 0:
        owners = sorted(c for cs in chunks for c in cs)
        q.put((ok, owners, sum(1 for c in chunks if not c), rounds))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T,B,idle", [(13, 2, 1), (10, 2, 3), (40, 2, 0)])
def test_result_exchange_async_world8_gloo_with_idle_ranks(T, B, idle):
    ok, owners, n_idle, rounds = _spawn(_strong_worker, 8, T, B)
    assert ok and owners == list(range(-(-T // B))) and n_idle == idle and rounds == -(-(-(-T // B)) // 8)


def test_list_chunks_is_the_gpu_free_twin_of_iter_chunks():
    """--gather: every rank derives every rank's chunk list from the shard plan alone (stream.list_chunks), so the merge rank knows
    which frames a gathered buffer holds; it must agree with what stream.iter_chunks hands the stages."""
    import numpy as np
    from premvos_amd import stream
    for n_frames, batch, world in ((13, 2, 8), (5, 8, 2), (16, 4, 3), (1, 2, 2)):
        images = [f"/v/{t:05d}.jpg" for t in range(n_frames)]
        for rank in range(world):
            for _, first, end in P.plan_shards([n_frames], world, rank, batch):
                a = stream.list_chunks(images, first, end, batch)
                b = [(names, nxt is not None) for names, frames, nxt in
                     stream.iter_chunks(images, first, end, batch, lambda fn: np.zeros((2, 2, 3), np.uint8))]
                assert a == b, (n_frames, batch, world, rank)


def _shard8_worker(rank, world, port, q, counts, batch):
    """plan_shards + iter_chunks at world 8 on a multi-video tree (fewer videos than ranks: frame ranges, rotated start rank)."""
    import numpy as np
    from premvos_amd import stream
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    owned, pairs, sizes = [], [], []
    for v, first, end in P.plan_shards(counts, world, rank, batch):
        images = [f"/video{v}/{t:05d}.jpg" for t in range(counts[v])]
        for names, frames, nxt in stream.iter_chunks(images, first, end, batch, lambda fn: np.full((2, 2, 3), int(os.path.basename(fn)[:5]), np.int32)):
            ids = [int(f[0, 0, 0]) for f in frames]
            owned += [(v, t) for t in ids]
            sizes.append((v, len(ids)))
            second = ids[1:] + ([int(nxt[0, 0, 0])] if nxt is not None else [])
            pairs += [(v, a, b) for a, b in zip(ids, second)]
    own = [None] * world
    dist.all_gather_object(own, (owned, pairs, sizes))
    if rank == 0:
        q.put(own)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("counts,batch", [([13, 4, 7], 2), ([5], 8), ([70, 50, 30, 90, 20, 11, 64, 8, 3], 8)])
def test_sharding_world8_gloo_ragged_videos(counts, batch):
    own = _spawn(_shard8_worker, 8, counts, batch)
    assert sorted(x for o, _, _ in own for x in o) == [(v, t) for v, n in enumerate(counts) for t in range(n)]
    assert sorted(x for _, p, _ in own for x in p) == [(v, t, t + 1) for v, n in enumerate(counts) for t in range(n - 1)]
    if len(counts) < 8:      # frame ranges: the chunks are exactly the ones a single rank forms
        assert sorted(s for _, _, ss in own for s in ss) == sorted((v, c) for v, n in enumerate(counts)
                                                                   for c in [batch] * (n // batch) + [n % batch] * bool(n % batch))
    else:                    # whole videos per rank
        assert all(len({v for v, _ in o}) >= 1 for o, _, _ in own) and sum(1 for o, _, _ in own if len({v for v, _ in o}) == 2) == 1


def test_host_budget_is_sized_from_the_threads_per_rank():
    from premvos_amd import io_pipeline as iop
    assert iop.host_budget(256, 8) == {"threads_per_rank": 32, "decode": 4, "refine_lanes": 2, "total": 12}
    assert iop.host_budget(32, 8)["decode"] == 1 and iop.host_budget(32, 8)["refine_lanes"] == 1
    for cpus, world in ((256, 8), (64, 8), (16, 8), (8, 1)):
        b = iop.host_budget(cpus, world)
        assert b["total"] * world <= max(2 * cpus, 8 * world)          # never more than 2x the hardware threads (or the 8-thread floor)
