"""The shipped conv configuration table and the closed-form rule behind it (premvos_amd.ops), without a GPU: the table parses,
its entries are legal candidates of their signatures, and the rule is a pure function of the signature."""
import json
import os

from premvos_amd import ops
from premvos_amd._lib import ConvDesc

NAMES = "n h w cin ho wo cout kh kw sh sw dh dw res out_mode precision in_ps out_ps w2 w4".split()


def _desc(sig):
    d, kv = ConvDesc(), dict(zip(NAMES, sig))
    for f in "n h w cin ho wo cout kh kw sh sw dh dw out_mode precision in_ps out_ps".split():
        setattr(d, f, kv[f])
    d.res = 1 if kv["res"] else None
    d.wgt_wino, d.wgt_wino4 = (1 if kv["w2"] else None), (1 if kv["w4"] else None)
    d.cin_pad, d.cout_pad = (d.cin + 3) // 4 * 4, (d.cout + 31) // 32 * 32
    d.k_pad = (d.kh * d.kw * d.cin_pad + 15) // 16 * 16
    same = (d.ho, d.wo) == (d.h, d.w)
    d.pt = d.pl = d.dh * (d.kh // 2) if (same or d.sh > 1) else 0
    return d


def test_shipped_table_parses_and_holds_legal_configurations():
    assert os.path.exists(ops.TUNE_TABLE)
    table = json.load(open(ops.TUNE_TABLE))
    assert len(table) > 100 and table == sorted(table)                      # sorted: the same choices give the same file hash
    fams = set()
    for sig, choice in table:
        assert len(sig) == len(NAMES) and len(choice) == 5
        d = _desc(sig)
        assert ops._sig(d) == tuple(sig)
        hint = choice[0]
        fams.add(hint if hint < 16 else 0)
        if hint >= 16:
            assert (hint >> 16, hint & 0xFFFF) in {(256, 128), (256, 129), (128, 128), (128, 96), (128, 64), (128, 32), (64, 128), (64, 64), (64, 32)}
            assert choice[1] in (16, 32)
            if hint & 0xFFFF == 129:                                         # (256 x 128 with four 128 x 64 waves: plain form, pointwise layers)
                assert tuple(choice[1:]) == (16, -1, 0, 0) and (d.kh, d.kw) == (1, 1) and d.cout >= 128
        ops.numerics_key(d, tuple(choice))                                   # defined for every entry
        if hint == 5:                                                        # an order-neutral stand-in for the implicit GEMM
            assert ops.numerics_key(d, tuple(choice)) == (0, None, None) and d.k_pad in (64, 128) and d.cout % 128 == 0
        if hint == 6:                                                        # ... and so is the LDS-DMA staged pointwise kernel (round 5)
            assert ops.numerics_key(d, tuple(choice)) == (0, None, None) and (d.kh, d.kw, d.pt, d.pl) == (1, 1, 0, 0) and d.k_pad >= 32
    assert fams == {0, 1, 2, 3, 4, 5, 6}        # implicit GEMM, direct, F(2x2) slab / slab-free, F(4x4), short-K streaming / LDS-DMA pointwise


def test_rule_choice_is_a_function_of_the_signature_only():
    table = json.load(open(ops.TUNE_TABLE))
    agree = 0
    for sig, choice in table:
        d = _desc(sig)
        r1, r2 = ops.rule_choice(d), ops.rule_choice(_desc(sig))
        assert r1 == r2 and len(r1) == 5
        assert r1 in ops._candidates(d) or r1[0] >= 16                      # a Winograd / direct choice is always a listed candidate
        agree += ops.numerics_key(d, r1) == ops.numerics_key(d, tuple(choice))
    assert agree >= 0.5 * len(table)                                        # the rule restates the regularities of the measured table


def test_numerics_key_separates_what_changes_the_summation_order():
    sig = [4, 46, 83, 1024, 46, 83, 512, 1, 1, 1, 1, 1, 1, False, 0, 0, 1024, 512, False, False]
    d = _desc(sig)
    t128, t64 = (128 << 16) | 128, (64 << 16) | 64
    assert ops.numerics_key(d, (t128, 16, -1, 0, 0)) == ops.numerics_key(d, (t64, 32, -1, 0, 0))          # tile, stage depth: neutral
    assert ops.numerics_key(d, (t128, 16, 2, 0, 0)) != ops.numerics_key(d, (t128, 16, -1, 0, 0))          # k-slices are not
    assert ops.numerics_key(d, (t128, 16, 2, 0, 0)) == ops.numerics_key(d, (t64, 32, 2, 0, 0))            # 1024 = 2 x 512 either way
    assert ops.numerics_key(d, (t128, 16, 4, 0, 0)) != ops.numerics_key(d, (t128, 16, 2, 0, 0))
    assert ops.numerics_key(d, (t128, 16, -1, 3, 4)) != ops.numerics_key(d, (t128, 16, -1, 0, 0))         # a tail split re-orders its rows
    assert ops.numerics_key(d, (2, 0, -1, 0, 0)) == ops.numerics_key(d, (2, 64, -1, 0, 0)) != ops.numerics_key(d, (3, 0, -1, 0, 0))


def test_kept_winograd_slab_plan_of_a_densenet_level():
    """ops.wino4_slab_plan (host logic only): PWC-Net level 3's F(4x4) layers -- windows [448, 597), [320, 597), [192, 597),
    [96, 597) of one concat buffer (PWCNet.py:201-205) -- share a slab of pitch 608; the first transforms its whole Kp, the others
    only what the previous layer prepended.  Shapes the shared zero padding would not cover get no plan."""
    from premvos_amd import ops
    from premvos_amd._lib import ConvDesc

    def desc(off, hint=4, end=597, n=16, h=64, w=112):
        d = ConvDesc()
        d.n, d.h, d.w, d.ho, d.wo, d.cin, d.tile_hint, d.dh, d.dw = n, h, w, h, w, end - off, hint, 1, 1
        d.cin_pad = (d.cin + 3) // 4 * 4
        return d, off

    need, plan = ops.wino4_slab_plan([desc(448), desc(320), desc(192), desc(96)])
    assert [(c0, t) for _, _, c0, t in plan] == [(448, 160), (320, 128), (192, 128), (96, 96)]
    assert {p for _, p, _, _ in plan} == {608} and need == 36 * 16 * 16 * 28 * 608
    assert ops.wino4_slab_plan([desc(448), desc(320, hint=2)]) is None            # a layer on another kernel
    assert ops.wino4_slab_plan([desc(448), desc(324)]) is None                    # window not at a multiple of 16
    assert ops.wino4_slab_plan([desc(448), desc(320, end=600)]) is None           # windows that end at different channels
    assert ops.wino4_slab_plan([desc(448), desc(320, h=32)]) is None              # different maps
    assert ops.wino4_slab_plan([]) is None
    # a later layer whose window starts ABOVE what is already in the slab transforms nothing
    _, plan = ops.wino4_slab_plan([desc(320), desc(448)])
    assert [(c0, t) for _, _, c0, t in plan] == [(320, 288), (448, 0)]
