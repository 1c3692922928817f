"""Where conv configurations come from (premvos_amd.ops.autotune): the shipped table, or -- for a signature it does not hold --
a closed-form rule for everything that changes the ORDER of the fp32 sums plus wall-clock timing of the order-neutral knobs.
The claims tested here: (1) configurations with the same ``numerics_key`` really are bit-identical, (2) the rule's choice is a
legal configuration for every kind of layer the nets hold, (3) two tuning passes of one process-independent layer freeze
configurations with the same arithmetic."""
import pytest
import torch

pytestmark = pytest.mark.gpu

LAYERS = [  # n, cin, cout, h, w, k, stride, dil
    (2, 728, 728, 25, 25, 1, 1, 1),       # Xception middle-flow pointwise: ragged last column tile
    (1, 256, 256, 46, 83, 3, 1, 1),       # ResNet conv4 3x3: implicit GEMM, F(2x2) slab / slab-free, F(4x4)
    (4, 81, 128, 8, 14, 3, 1, 1),         # coarse PWC level: k-splits
    (1, 1024, 512, 14, 14, 1, 1, 1),      # few tiles, long K
    (2, 128, 128, 32, 40, 3, 1, 2),       # atrous context layer (slab-free Winograd only)
    (1, 64, 96, 64, 112, 3, 2, 1),        # stride 2
    (2, 565, 2, 32, 56, 3, 1, 1),         # flow head: direct small-N kernel
    (2, 64, 168, 30, 30, 1, 1, 1),        # last column tile holds 40 real columns: the 128x128 tile's narrow (4x1 waves) path, 2 blocks
    (1, 96, 148, 40, 40, 1, 1, 1),        # ... 20 real columns: 1 block
]


def _layer(ops, n, cin, cout, h, w, k, stride, dil, seed=0):
    g = torch.Generator().manual_seed(seed + cin + h)
    x = ops.NHWC.alloc(n, h, w, cin)
    x.buf.copy_(torch.randn(x.buf.shape, generator=g))
    wt = torch.randn((cout, cin, k, k), generator=g) * (2.0 / (k * k * cin)) ** 0.5
    pk = ops.pack_conv(wt, torch.randn((cout,), generator=g) * 0.1)
    pad = dil * (k // 2)
    ho, wo = ops.out_size(h, k, stride, pad, pad, dil), ops.out_size(w, k, stride, pad, pad, dil)
    out = ops.NHWC.alloc(n, ho, wo, cout)
    d = ops.conv_desc(x, pk, out, stride=(stride, stride), dilation=(dil, dil), pad=(pad, pad), act=ops.ACT_RELU)
    return x, pk, out, d


def _run(ops, d, out, cand, ws):
    d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = cand
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    out.buf.fill_(-7.0)
    ops.run_desc(d)
    torch.cuda.synchronize()
    return out.buf.clone()


@pytest.mark.parametrize("layer", LAYERS)
def test_order_neutral_knobs_are_bit_identical(layer):
    from premvos_amd import ops
    x, pk, out, d = _layer(ops, *layer)
    cands = ops._candidates(d) + [ops.rule_choice(d)]
    need = 0
    for c in cands:
        d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = c
        need = max(need, ops.workspace_bytes(d))
    ws = torch.empty(need // 4 + 1, dtype=torch.float32, device="cuda")
    groups = {}
    for c in cands:
        groups.setdefault(ops.numerics_key(d, c), []).append(c)
    assert len(groups) >= 1 and sum(len(v) for v in groups.values()) == len(cands)
    results = {}
    for key, members in groups.items():
        ref = _run(ops, d, out, members[0], ws)
        assert torch.isfinite(ref).all() and not (ref == -7.0).all()
        for c in members[1:]:
            assert torch.equal(_run(ops, d, out, c, ws), ref), (key, members[0], c)
        results[key] = ref
    # different keys = a different order of the same sum: close, and (at least somewhere) not the same bits
    base = next(iter(results.values()))
    scale = max(1.0, float(base.abs().max()))
    for key, r in results.items():
        assert float((r - base).abs().max()) < 2e-4 * scale, key


@pytest.mark.parametrize("layer", LAYERS)
def test_rule_choice_is_deterministic_and_two_tuning_passes_agree(layer, monkeypatch):
    from premvos_amd import ops
    monkeypatch.setenv("PREMVOS_TUNE_TABLE", "0")
    monkeypatch.delenv("PREMVOS_TUNE_CACHE", raising=False)
    monkeypatch.setenv("PREMVOS_AUTOTUNE", "1")
    x, pk, out, d = _layer(ops, *layer)
    saved, state = dict(ops._TUNE_CACHE), dict(ops._TUNE_STATE)
    try:
        keys, outs = [], []
        for _ in range(2):
            ops._TUNE_CACHE.clear()
            ops._TUNE_STATE.update(loaded=True)
            ops.autotune([d])
            cand = (d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k)
            keys.append(ops.numerics_key(d, cand))
            ws = ops.assign_workspace([d])          # noqa: F841
            out.buf.fill_(0.0)
            ops.run_desc(d)
            torch.cuda.synchronize()
            outs.append(out.buf.clone())
        assert keys[0] == keys[1] == ops.numerics_key(d, ops.rule_choice(d))
        assert torch.equal(outs[0], outs[1])        # whatever tile the stopwatch preferred, the bits are the same
    finally:
        ops._TUNE_CACHE.clear()
        ops._TUNE_CACHE.update(saved)
        ops._TUNE_STATE.update(state)


def test_tuner_compares_outputs_before_it_times(monkeypatch):
    """`_time_cands` chose by time alone.  Now every candidate's output window is digested on the GPU
    (premvos_digest_u64) and a candidate that shares its numerics key with an earlier one but wrote other bits is dropped."""
    import ctypes as C
    import warnings
    from premvos_amd import _lib, ops
    lib, st = _lib.load(), _lib.current_stream()
    g = torch.Generator().manual_seed(3)
    x = ops.NHWC.alloc(1, 40, 41, 128)
    x.buf.copy_(torch.randn(x.buf.shape, generator=g))
    pk = ops.pack_conv(torch.randn((256, 128, 1, 1), generator=g) * 0.1, torch.randn((256,), generator=g))
    out = ops.NHWC.alloc(1, 40, 41, 256)
    d = ops.conv_desc(x, pk, out, act=ops.ACT_RELU)
    cands = [((128 << 16) | 128, 16, -1, 0, 0), (5, 0, -1, 0, 0), ((64 << 16) | 64, 16, -1, 0, 0)]
    d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = cands[0]
    ops.run_desc(d)
    a = ops._out_digest(d, lib, st)
    assert a == ops._out_digest(d, lib, st) and a != 0
    out.buf[0, 7, 9, 3] += 1.0
    assert ops._out_digest(d, lib, st) != a                                   # one word changed: another digest
    assert ops._time_cands(d, cands, lib, st, 2) in cands                     # bit-identical candidates: all kept, one wins
    # a "candidate" of the same numerics class that writes other bits: simulated by a library call that scribbles afterwards
    real = lib.premvos_conv2d_f32
    calls = {"n": 0}

    class Scribbler:
        def __call__(self, dref, stream):
            rc = real(dref, stream)
            calls["n"] += 1
            if d.tile_hint == 5:
                out.buf[0, 0, 0, 0] = 12345.0
            return rc
    monkeypatch.setattr(lib, "premvos_conv2d_f32", Scribbler(), raising=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        best = ops._time_cands(d, cands, lib, st, 2)
    assert best in (cands[0], cands[2]) and any("does not reproduce" in str(m.message) for m in w)
