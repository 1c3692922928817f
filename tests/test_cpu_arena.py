"""premvos_amd.arena: liveness-planned activation memory of a launch plan (host logic; the GPU side is
tests/test_gpu_arena.py)."""
import random

import pytest
import torch

from premvos_amd import arena


def _check(items, offs, total):
    al = lambda b: (b + arena.ALIGN - 1) // arena.ALIGN * arena.ALIGN      # noqa: E731
    for i, (b, a, e) in enumerate(items):
        assert offs[i] % arena.ALIGN == 0 and offs[i] + al(b) <= total
        for j in range(i):
            bj, aj, ej = items[j]
            if aj < e and a < ej:                                          # live at the same time: disjoint bytes
                assert offs[i] + al(b) <= offs[j] or offs[j] + al(bj) <= offs[i], (i, j)


def test_pack_intervals_never_overlaps_live_tensors_and_reaches_the_chain_bound():
    rng = random.Random(7)
    for _ in range(50):
        items = []
        for _ in range(rng.randint(1, 60)):
            a = rng.randint(0, 100)
            items.append((rng.randint(1, 1 << 20), a, a + rng.randint(1, 30)))
        offs, total = arena.pack_intervals(items)
        _check(items, offs, total)
        assert total <= sum((b + 255) // 256 * 256 for b, _, _ in items)
    # a ResNet-like chain x -> t1 -> t2 -> y (y reads x as residual): four tensors live at most, whatever the depth
    items, t = [], 0
    big, small = 4 << 20, 1 << 20
    x_alloc = 0
    for blk in range(23):
        items.append((small, t + 1, t + 7))        # t1
        items.append((small, t + 2, t + 7))        # t2
        items.append((big, t + 3, t + 17))         # y: live until the next block's conv3 has read it
        t += 10
    offs, total = arena.pack_intervals(items)
    _check(items, offs, total)
    assert total <= 2 * big + 2 * small + 4 * arena.ALIGN


def test_two_pass_builder_reuses_bytes_and_keeps_unreleased_tensors(monkeypatch):
    seen = {}

    def build(A):
        x = A.alloc(2, 8, 8, 16)
        pad = A.alloc(2, 8, 8, 4, pooled=False)                # a tensor with padding channels: outside the arena
        chain = []
        for i in range(6):
            y = A.alloc(2, 8, 8, 16)
            chain.append(y)
            A.release(x)
            x = y
        named = A.alloc(1, 4, 4, 8)                            # never released: live to the end
        seen[A.dry] = (chain, named, pad, x)

    A = arena.two_pass("cpu", build)
    rep = A.report()
    one = 4 * 2 * 8 * 8 * 16
    assert rep["tensors"] == 8 and rep["one_buffer_per_tensor_bytes"] == 7 * one + 4 * 16 * 8
    assert rep["arena_bytes"] <= 2 * one + 4 * 16 * 8 + 3 * arena.ALIGN          # two chain tensors live at a time + the named one
    assert rep["arena_bytes"] >= rep["peak_live_bytes"]
    chain, named, pad, last = seen[False]
    assert all(t.device.type == "cpu" and t.shape == (2, 8, 8, 16) and float(t.abs().sum()) == 0.0 for t in chain)
    assert chain[0].data_ptr() == chain[2].data_ptr() == chain[4].data_ptr() != chain[1].data_ptr()      # rotation
    lo, hi = named.data_ptr(), named.data_ptr() + 4 * named.numel()
    l2, h2 = last.data_ptr(), last.data_ptr() + 4 * last.numel()
    assert hi <= l2 or h2 <= lo                                                   # both live at the end: disjoint
    assert seen[True][0][0].device.type == "meta"
    # switched off: one tensor per alloc, release is a no-op
    monkeypatch.setenv("PREMVOS_ARENA", "0")
    B = arena.two_pass("cpu", build)
    chain = seen[False][0]
    assert len({t.data_ptr() for t in chain}) == 6 and B.report()["arena_bytes"] == 0


def test_diverging_passes_are_caught():
    n = {"k": 0}

    def build(A):
        n["k"] += 1
        A.alloc(1, 2, 2, 4 * n["k"])

    with pytest.raises(AssertionError):
        arena.two_pass("cpu", build)


def test_plans_of_one_lane_share_their_bytes():
    def builder(n):
        def build(A):
            x = A.alloc(n, 4, 4, 8)
            y = A.alloc(n, 4, 4, 8)
            A.release(x)
            build.out = (x, y)
        return build

    lane = {}
    b8, b4, b16 = builder(8), builder(4), builder(16)
    a8 = arena.two_pass("cpu", b8, shared=lane)
    big = lane["f32"]
    a4 = arena.two_pass("cpu", b4, shared=lane)                 # a smaller plan of the lane lives in the same bytes
    assert lane["f32"] is big and a4.bufs["f32"] is big and b4.out[0].data_ptr() == b8.out[0].data_ptr()
    a16 = arena.two_pass("cpu", b16, shared=lane)               # a larger one replaces the lane's buffer; the earlier plans keep theirs
    assert lane["f32"] is a16.bufs["f32"] and a16.bufs["f32"] is not big and a8.bufs["f32"] is big
    assert a16.bufs["f32"].numel() * 4 >= a16.report()["peak_live_bytes"]


def test_release_refuses_channel_windows_views_and_unknown_tensors():
    """``NHWC.slice`` shares its parent's storage, so releasing a window would end the whole buffer's lifetime; an
    image range is a different tensor object and used to be ignored silently."""
    from premvos_amd.ops import NHWC

    def build(A):
        cat = NHWC(A.alloc(2, 4, 4, 32))
        A.release(cat.slice(0, 32))                            # the full range of the buffer: the same thing as the tensor itself

    arena.two_pass("cpu", build)
    for bad in ("window", "offset-window", "images", "foreign", "s8-window"):
        def build(A, bad=bad):
            cat = NHWC(A.alloc(2, 4, 4, 32))
            s8 = NHWC(A.alloc(2, 4, 4, 32, "s8"), layout="s8")
            A.release({"window": lambda: cat.slice(0, 16), "offset-window": lambda: cat.slice(16, 16),
                       "images": lambda: cat.images(0, 1), "foreign": lambda: torch.zeros(1, 1, 1, 4),
                       "s8-window": lambda: s8.slice(8, 8)}[bad]())
        with pytest.raises(ValueError):
            arena.two_pass("cpu", build)


def test_side_tensors_take_no_memory_in_the_dry_pass():
    kinds = []

    def build(A):
        t = A.side((3, 5), torch.int32)
        kinds.append((A.dry, t.device.type, tuple(t.shape), t.dtype))
        u = A.side(7, torch.float32)
        assert tuple(u.shape) == (7,)
        if not A.dry:
            assert int(t.abs().sum()) == 0

    arena.two_pass("cpu", build)
    assert kinds == [(True, "meta", (3, 5), torch.int32), (False, "cpu", (3, 5), torch.int32)]
