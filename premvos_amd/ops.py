"""Thin Python wrappers over the C-ABI: NHWC slice views, weight packing, one function per kernel.

PyTorch is used for device memory and streams only; every computation below is a call into
libpremvos_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, OUT_NHWC, OUT_PIXSHUF2, ConvDesc  # noqa: F401


def _r(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class NHWC:
    """A channel window [coff, coff+c) of a pixel-major fp32 buffer [n,h,w,ps]."""

    __slots__ = ("buf", "n", "h", "w", "c", "ps", "coff", "layout")

    def __init__(self, buf: torch.Tensor, c: Optional[int] = None, coff: int = 0, layout: str = "f32"):
        assert buf.dim() == 4 and buf.dtype == torch.float32 and buf.is_contiguous()
        self.buf = buf
        # "f32": floats.  "s8": the resident split layout of the bf16x3 mode (csrc/conv_bf16x3_s8.hip) -- every group of 8 channels is
        # the 32 bytes {hi(8 x bf16), lo(8 x bf16)} stored IN PLACE of its eight floats; readers that expect floats refuse such a buffer
        self.layout = layout
        self.n, self.h, self.w, self.ps = buf.shape
        self.coff = coff
        self.c = self.ps - coff if c is None else c
        assert 0 <= coff and coff + self.c <= self.ps

    @staticmethod
    def alloc(n: int, h: int, w: int, c: int, device=None, ps: Optional[int] = None) -> "NHWC":
        ps = _r(c, 4) if ps is None else ps
        return NHWC(torch.zeros((n, h, w, ps), dtype=torch.float32, device=_lib.resolve_device(device)), c=c)

    def slice(self, coff: int, c: int) -> "NHWC":
        assert self.layout == "f32" or (coff % 8 == 0 and c % 8 == 0), "an S8 window starts and ends on a group of 8 channels"
        return NHWC(self.buf, c=c, coff=self.coff + coff, layout=self.layout)

    def images(self, n0: int, n: int) -> "NHWC":
        return NHWC(self.buf[n0:n0 + n], c=self.c, coff=self.coff, layout=self.layout)

    @staticmethod
    def alloc_s8(n: int, h: int, w: int, c: int, device=None) -> "NHWC":
        v = NHWC(torch.zeros((n, h, w, _r(c, 8)), dtype=torch.float32, device=_lib.resolve_device(device)), c=c, layout="s8")
        return v

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr() + 4 * self.coff

    def torch(self) -> torch.Tensor:
        """NCHW copy (tests / debugging); an S8 buffer is decoded (hi + lo, the value its consumers multiply)."""
        if self.layout == "s8":
            g = self.buf[..., self.coff:self.coff + _r(self.c, 8)].contiguous().view(torch.bfloat16)      # [n,h,w,groups*16]
            g = g.view(self.n, self.h, self.w, -1, 2, 8).float()
            return (g[..., 0, :] + g[..., 1, :]).reshape(self.n, self.h, self.w, -1)[..., :self.c].permute(0, 3, 1, 2).contiguous()
        return self.buf[..., self.coff:self.coff + self.c].permute(0, 3, 1, 2).contiguous()


@dataclass
class PackedConv:
    """Device-resident packed weights of one conv (see premvos_hip.h for the layout)."""
    wgt: torch.Tensor
    bias: Optional[torch.Tensor]
    cin: int
    cout: int
    kh: int
    kw: int
    cin_pad: int
    k_pad: int
    cout_pad: int
    cout_ps: int = 0          # >0: transposed-conv phases (PIXSHUF2)
    precision: int = 0        # _lib.PREC_*; bf16 modes: wgt / wgt_lo are bfloat16 [cout_pad][k_pad], k_pad % 32 == 0
    wgt_lo: Optional[torch.Tensor] = None
    wgt_wino: Optional[torch.Tensor] = None      # fp32 3x3 layers: the 16 Winograd F(2x2,3x3) filter transforms [16][cout_pad][r16(cin_pad)]
    wgt_wino4: Optional[torch.Tensor] = None     # K-rich fp32 3x3 layers: the 36 F(4x4,3x3) filter transforms, same packing


def default_precision() -> str:
    import os
    return os.environ.get("PREMVOS_PRECISION", "fp32")


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor], device=None,
              scale: Optional[torch.Tensor] = None, precision: str = "fp32", wino4_min_c: Optional[int] = None) -> PackedConv:
    """OIHW fp32 -> [cout_pad][k_pad] with k = (kh*KW+kw)*cin_pad + c.  ``scale`` (per cout)
    folds a frozen BatchNorm's gamma/sqrt(var+eps) into the weights.  precision 'bf16' / 'bf16x3': the matrix
    is stored as bfloat16 high parts (+ low parts w - float(hi)) and k is padded to 32."""
    prec = _lib.PRECISIONS[precision]
    device = _lib.resolve_device(device)
    # (round 6: the raw tensor is uploaded ONCE and folded / permuted / padded on the device -- the same IEEE operations as on the
    #  host, so the packed bits do not change; packing the ~360 convs of the three nets on the host cost 3.5 s of a rank's cold start)
    w = weight.detach().to(device=device, dtype=torch.float32)
    cout, cin, kh, kw = w.shape
    if scale is not None:
        w = w * scale.detach().to(device=device, dtype=torch.float32).view(-1, 1, 1, 1)
    cin_pad, cout_pad = _r(cin, 4), _r(cout, 32)
    k = kh * kw * cin_pad
    k_pad = _r(k, 16 if prec == _lib.PREC_F32 else 32)
    p = torch.zeros((cout_pad, kh * kw, cin_pad), dtype=torch.float32, device=device)
    p[:cout, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    full = torch.zeros((cout_pad, k_pad), dtype=torch.float32, device=device)
    full[:, :k] = p.reshape(cout_pad, k)
    b = None
    if bias is not None:
        b = torch.zeros(cout_pad, dtype=torch.float32, device=device)
        b[:cout] = bias.detach().to(device=device, dtype=torch.float32)
    if prec == _lib.PREC_F32:
        pk = PackedConv(full, b, cin, cout, kh, kw, cin_pad, k_pad, cout_pad)
        if (kh, kw) == (3, 3) and cout % 4 == 0 and cin >= WINO_MIN_CIN and wino_enabled():
            pk.wgt_wino = pack_winograd(w, cin_pad, cout_pad, device)
            if cin >= (WINO4_MIN_C if wino4_min_c is None else wino4_min_c) and cout >= WINO4_MIN_COUT and wino4_enabled():
                pk.wgt_wino4 = pack_winograd4(w, cin_pad, cout_pad, device)
        return pk
    hi = full.to(torch.bfloat16)
    if prec != _lib.PREC_BF16X3:
        return PackedConv(hi.contiguous(), b, cin, cout, kh, kw, cin_pad, k_pad, cout_pad, 0, prec, None)
    both = torch.stack([hi, (full - hi.float()).to(torch.bfloat16)]).contiguous()
    return PackedConv(both[0], b, cin, cout, kh, kw, cin_pad, k_pad, cout_pad, 0, prec, both[1])


WINO_MIN_CIN = 16


def wino_enabled() -> bool:
    import os
    return os.environ.get("PREMVOS_WINOGRAD", "1") != "0"


WINO4_MAX_WS = 16 << 30     # bytes of workspace one layer may ask for
WINO4_MIN_C, WINO4_MIN_COUT = int(os.environ.get("PREMVOS_WINO4_MIN_C", "128")), int(os.environ.get("PREMVOS_WINO4_MIN_COUT", "32"))       # F(4x4,3x3) moves 2.25x the input and output through workspace slabs: only K- and N-rich layers gain


def wino4_enabled() -> bool:
    import os
    return os.environ.get("PREMVOS_WINOGRAD4", "1") != "0"


def pack_winograd4(w_oihw: torch.Tensor, cin_pad: int, cout_pad: int, device="cpu") -> torch.Tensor:
    """U[6*i+j] = (G g G^T)[i][j] of F(4x4,3x3) (Lavin & Gray 2016; points 0, +-1, +-2, inf) for every (cout, cin) filter g, in
    float64 on ``device`` and rounded once; packed like ``pack_winograd``: [36][cout_pad][roundup(cin_pad, 16)].
    G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]] written out as element-wise float64
    sums (IEEE: the same bits on host and GPU, and no BLAS call at model set-up)."""
    cout, cin = w_oihw.shape[:2]
    g = w_oihw.to(device=device, dtype=torch.float64)

    def G(a, b, c):                                  # the six rows of G applied to three values
        return (a / 4, -(a + b + c) / 6, -(a - b + c) / 6, (a + 2 * b + 4 * c) / 24, (a - 2 * b + 4 * c) / 24, c)
    out = torch.zeros((36, cout_pad, _r(cin_pad, 16)), dtype=torch.float32, device=device)
    for i, r in enumerate(G(g[:, :, 0], g[:, :, 1], g[:, :, 2])):             # filter rows -> [cout, cin, 3]
        for j, col in enumerate(G(r[..., 0], r[..., 1], r[..., 2])):
            out[6 * i + j, :cout, :cin] = col.to(torch.float32)
    return out


def pack_winograd(w_oihw: torch.Tensor, cin_pad: int, cout_pad: int, device="cpu") -> torch.Tensor:
    """U[4*xi+nu] = (G g G^T)[xi][nu] of every (cout, cin) filter g (BatchNorm scale already folded into ``w_oihw``), computed in
    float64 and rounded once; each packed like a 1x1 conv: [cout_pad][roundup(cin_pad, 16)], zero padded.
    G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] written out as element-wise float64 sums on ``device`` (IEEE, same bits on host
    and GPU; on the GPU the 3x3 layers of ResNet-101 pack in milliseconds instead of seconds of model set-up)."""
    cout, cin = w_oihw.shape[:2]
    g = w_oihw.to(device=device, dtype=torch.float64)
    r0, r1, r2 = g[:, :, 0], g[:, :, 1], g[:, :, 2]                          # filter rows [cout, cin, 3]
    out = torch.zeros((16, cout_pad, _r(cin_pad, 16)), dtype=torch.float32, device=device)
    for xi, r in enumerate((r0, 0.5 * (r0 + r1 + r2), 0.5 * (r0 - r1 + r2), r2)):
        c0, c1, c2 = r[..., 0], r[..., 1], r[..., 2]
        for nu, col in enumerate((c0, 0.5 * (c0 + c1 + c2), 0.5 * (c0 - c1 + c2), c2)):
            out[4 * xi + nu, :cout, :cin] = col.to(torch.float32)
    return out


def pack_deconv4x4s2(weight: torch.Tensor, bias: Optional[torch.Tensor], device=None,
                     precision: str = "fp32") -> PackedConv:
    """ConvTranspose2d(k=4,s=2,p=1) weights [cin,cout,4,4] -> the equivalent 3x3 conv with
    4*cout phase outputs (phase = 2*py+px writes out[2y+py][2x+px]).  For output row 2y+py the
    contributing input rows are y+dy with  py=0: (dy=0,ky=1),(dy=-1,ky=3);  py=1: (dy=0,ky=2),(dy=+1,ky=0)."""
    w = weight.detach().to(torch.float32).cpu()
    cin, cout, kh, kw = w.shape
    assert (kh, kw) == (4, 4)
    taps = {0: ((0, 1), (-1, 3)), 1: ((0, 2), (1, 0))}
    w3 = torch.zeros((4 * cout, cin, 3, 3), dtype=torch.float32)
    for py in (0, 1):
        for px in (0, 1):
            ph = 2 * py + px
            for dy, ky in taps[py]:
                for dx, kx in taps[px]:
                    w3[ph * cout:(ph + 1) * cout, :, dy + 1, dx + 1] = w[:, :, ky, kx].t()
    b4 = None if bias is None else bias.detach().to(torch.float32).cpu().repeat(4)
    pk = pack_conv(w3, b4, device, precision=precision)
    pk.cout_ps = cout
    return pk


def pack_deconv2x2s2(weight: torch.Tensor, bias: Optional[torch.Tensor], device=None,
                     precision: str = "fp32") -> PackedConv:
    """ConvTranspose2d(k=2,s=2) (tensorpack Deconv2D(…, 2, stride=2), proposal_net/model.py:507): the taps do not
    overlap, so it is a 1x1 conv with 4*cout phase outputs (phase = 2*ky+kx writes out[2y+ky][2x+kx])."""
    w = weight.detach().to(torch.float32).cpu()
    cin, cout, kh, kw = w.shape
    assert (kh, kw) == (2, 2)
    w1 = w.permute(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1).contiguous()
    b4 = None if bias is None else bias.detach().to(torch.float32).cpu().repeat(4)
    pk = pack_conv(w1, b4, device, precision=precision)
    pk.cout_ps = cout
    return pk


def conv_desc(x: NHWC, pk: PackedConv, out: NHWC, stride=(1, 1), dilation=(1, 1), pad=(0, 0),
              act=ACT_NONE, slope=0.1, res: Optional[NHWC] = None, tile_hint: int = 0, split_k: int = 0,
              stage_k: int = 0) -> ConvDesc:
    """Build the descriptor (validated again on the C side).  ``pad`` = (top, left); the output
    size comes from ``out`` so asymmetric bottom/right padding is implicit."""
    assert x.c == pk.cin, (x.c, pk.cin)
    assert x.layout == "f32" and out.layout == "f32" and (res is None or res.layout == "f32"), "fp32 kernels read and write floats (S8 buffers: conv_s8)"
    d = ConvDesc()
    d.inp, d.wgt = x.ptr, pk.wgt.data_ptr()
    d.bias = pk.bias.data_ptr() if pk.bias is not None else None
    d.res = res.ptr if res is not None else None
    d.out = out.ptr
    d.n, d.h, d.w, d.cin, d.in_ps = x.n, x.h, x.w, x.c, x.ps
    if pk.cout_ps:
        assert out.c == pk.cout_ps and out.h == 2 * x.h and out.w == 2 * x.w and out.n == x.n
        d.ho, d.wo = x.h, x.w
        d.out_mode, d.cout_ps = OUT_PIXSHUF2, pk.cout_ps
    else:
        assert out.c == pk.cout and out.n == x.n, (out.c, pk.cout)
        d.ho, d.wo = out.h, out.w
        d.out_mode, d.cout_ps = OUT_NHWC, 0
    d.cout, d.out_ps = pk.cout, out.ps
    d.res_ps = res.ps if res is not None else 0
    d.kh, d.kw = pk.kh, pk.kw
    d.sh, d.sw = stride
    d.dh, d.dw = dilation
    d.pt, d.pl = pad
    d.cin_pad, d.k_pad, d.cout_pad = pk.cin_pad, pk.k_pad, pk.cout_pad
    d.act, d.slope = act, slope
    d.tile_hint = tile_hint
    d.split_k = split_k
    d.stage_k = stage_k
    d.tail_m_tiles, d.tail_split_k = 0, 0
    d.workspace, d.workspace_bytes = None, 0
    d.precision = pk.precision
    d.wgt_lo = pk.wgt_lo.data_ptr() if pk.wgt_lo is not None else None
    d.wgt_wino = pk.wgt_wino.data_ptr() if pk.wgt_wino is not None else None
    d.wgt_wino4 = pk.wgt_wino4.data_ptr() if pk.wgt_wino4 is not None else None
    return d


def algorithmic_bytes(d: ConvDesc) -> float:
    """Compulsory HBM bytes of one conv launch: every input pixel the taps touch once, weights once, output (+ residual)
    once -- what the PMC traffic figure of bench.py's roofline is compared with."""
    m = d.n * d.ho * d.wo
    pix_in = min(d.n * d.h * d.w, m * d.kh * d.kw)
    return 4.0 * (pix_in * d.cin + m * d.cout * (2 if d.res else 1) + d.cout * d.kh * d.kw * d.cin)


def workspace_bytes(d: ConvDesc) -> int:
    return int(_lib.load().premvos_conv2d_workspace_bytes(C.byref(d)))


def assign_workspace(descs, device=None) -> Optional[torch.Tensor]:
    """One split-K scratch buffer shared by all convs of a (stream-ordered) launch list."""
    device = _lib.resolve_device(device)
    need = max([workspace_bytes(d) for d in descs] + [0])
    if need == 0:
        return None
    ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
    for d in descs:
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    return ws


_TUNE_CACHE: dict = {}
# Building a launch plan (buffers, autotuning on the real buffers, HIP-graph capture) is serialised across host threads: the
# streaming driver and the IO lanes of the stage drivers run several nets from several threads.
BUILD_LOCK = __import__("threading").RLock()


def _sig(d: ConvDesc):
    return (d.n, d.h, d.w, d.cin, d.ho, d.wo, d.cout, d.kh, d.kw, d.sh, d.sw, d.dh, d.dw, bool(d.res), d.out_mode,
            d.precision, d.in_ps, d.out_ps, bool(d.wgt_wino), bool(d.wgt_wino4))


def wino_applicable(d: ConvDesc) -> bool:
    return bool(d.wgt_wino) and d.precision == _lib.PREC_F32 and (d.kh, d.kw, d.sh, d.sw, d.dh, d.dw) == (3, 3, 1, 1, 1, 1) \
        and d.out_mode == OUT_NHWC and d.cout % 4 == 0 and d.ho == d.h + 2 * d.pt - 2 and d.wo == d.w + 2 * d.pl - 2


def wino_atrous_applicable(d: ConvDesc) -> bool:
    """3x3 / stride 1 / dilation r / pad r ('SAME'): Winograd on the r x r sub-lattices (slab-free kernel only)."""
    return bool(d.wgt_wino) and d.precision == _lib.PREC_F32 and (d.kh, d.kw, d.sh, d.sw) == (3, 3, 1, 1) and d.dh == d.dw > 1 \
        and d.out_mode == OUT_NHWC and d.cout % 4 == 0 and (d.pt, d.pl) == (d.dh, d.dw) and (d.ho, d.wo) == (d.h, d.w)


def stream_applicable(d: ConvDesc) -> bool:
    """Mirror of premvos::conv_stream_applicable: 1x1 / stride 1 fp32 layers with cin = 64 or 128 (= k_pad) and cout % 128 == 0."""
    return d.precision == _lib.PREC_F32 and (d.kh, d.kw, d.sh, d.sw, d.pt, d.pl) == (1, 1, 1, 1, 0, 0) and d.ho == d.h and d.wo == d.w \
        and d.out_mode == OUT_NHWC and d.k_pad in (64, 128) and d.cin_pad == d.k_pad and d.cout % 128 == 0 and d.cout <= 512 \
        and d.in_ps % 4 == 0 and d.out_ps % 4 == 0 and (not d.res or d.res_ps % 4 == 0) and d.act in (ACT_NONE, ACT_RELU, ACT_LEAKY) \
        and (d.inp or 0) % 16 == 0 and (d.out or 0) % 16 == 0 and (d.res or 0) % 16 == 0


def pwdma_applicable(d: ConvDesc) -> bool:
    """Mirror of premvos::conv_pwdma_applicable (csrc/conv_pwdma_f32.hip): 1x1 fp32 layers without padding, K of two stages or more,
    the wide epilogue's alignment conditions."""
    return d.precision == _lib.PREC_F32 and (d.kh, d.kw, d.pt, d.pl) == (1, 1, 0, 0) and d.out_mode == OUT_NHWC \
        and d.ho == (d.h - 1) // d.sh + 1 and d.wo == (d.w - 1) // d.sw + 1 and d.k_pad >= 32 and d.k_pad % 16 == 0 and d.cin_pad % 4 == 0 \
        and d.cin_pad <= d.k_pad and d.k_pad - d.cin_pad < 16 and d.in_ps >= d.cin_pad and d.cout % 4 == 0 and d.out_ps % 4 == 0 \
        and d.in_ps % 4 == 0 and (d.inp or 0) % 16 == 0 and (d.out or 0) % 16 == 0 and (d.res or 0) % 16 == 0 \
        and (not d.res or d.res_ps % 4 == 0) and (d.bias or 0) % 16 == 0


def _candidates(d: ConvDesc):
    m = d.n * d.ho * d.wo
    if d.cout <= 32:
        tiles = [(128, 32), (64, 32)]
    elif d.cout <= 64:
        tiles = [(128, 64), (64, 64)]
    else:
        tiles = [(128, 128), (64, 128), (128, 64), (64, 64)]
        if d.precision == _lib.PREC_F32 and m >= 256 * 512 and os.environ.get("PREMVOS_TILE256", "1") != "0":
            tiles.append((256, 128))           # 8 waves of 64x64 at <= 128 VGPRs: four waves per SIMD, half the B staging per MFMA
        if d.precision == _lib.PREC_F32 and m >= 256 * 64 and d.cout >= 128 and (d.kh, d.kw) == (1, 1):
            tiles.append((256, 129))           # four waves of 128x64 (round 5; pointwise layers: tools/retile_pointwise.py)
        if d.precision == _lib.PREC_F32 and 64 < d.cout <= 96:
            tiles.append((128, 96))
    out = []
    if d.precision == _lib.PREC_F32 and d.cout <= 2 and d.out_mode == OUT_NHWC and d.kh * d.kw * d.cin_pad >= 32 \
            and d.cout * d.k_pad * 4 <= 150 * 1024:
        out.append((1, 0, -1, 0, 0))               # tile_hint 1 = the direct (non-MFMA) kernel for 1-2 output channels
    if wino_applicable(d):
        out.append((2, 0, -1, 0, 0))               # tile_hint 2 = Winograd F(2x2,3x3) (csrc/conv_wino_f32.hip), 128 tile rows
        if d.cout > 32:
            out.append((2, 64, -1, 0, 0))          # ... with 64-tile-row workgroups
    if wino_applicable(d) or wino_atrous_applicable(d):
        # tile_hint 3 = the same algebra without slabs: one kernel, a workgroup walks all 16 components of its block
        # (stage_k = block id, see conv_wino_f32.hip: tile rows x couts / waves / stage depth); it also takes atrous layers
        # (the slab-free kernel addresses its operands with 32-bit element offsets: csrc/conv_wino_f32.hip refuses larger tensors)
        small = d.n * d.h * d.w * d.in_ps < (1 << 30) and 16 * d.cout_pad * _r(d.cin_pad, 16) < (1 << 30)
        if small and os.environ.get("PREMVOS_WINOGRAD_FUSED", "1") != "0":
            out.extend((3, v, -1, 0, 0) for v in (() if d.cout <= 32 else (3, 5) if d.cout <= 64 else (0, 2, 4, 6)))
    if (wino_applicable(d) or wino_atrous_applicable(d)) and bool(d.wgt_wino4) and os.environ.get("PREMVOS_WINOGRAD4", "1") != "0":
        # tile_hint 4 = Winograd F(4x4,3x3) (csrc/conv_wino4_f32.hip): 4x fewer multiplies, 2.25x the input + output through slabs
        mt4 = d.n * d.dh * d.dw * -(-(-(-d.ho // d.dh)) // 4) * -(-(-(-d.wo // d.dw)) // 4)      # (atrous: per sub-lattice)
        if 36 * mt4 * (_r(d.cin_pad, 16) + _r(d.cout, 128)) * 4 <= WINO4_MAX_WS:
            out.extend((4, v, -1, 0, 0) for v in (0, 64, 16, 80))     # GEMM block: 128 / 64 tile rows x 32- / 16-deep stages
    if stream_applicable(d):
        out.append((5, 0, -1, 0, 0))               # tile_hint 5 = short-K streaming pointwise kernel (csrc/conv_stream_f32.hip): same sums
    if pwdma_applicable(d) and d.cout > 64 and os.environ.get("PREMVOS_PWDMA", "1") != "0":
        out.append((6, 0, -1, 0, 0))               # tile_hint 6 = LDS-DMA staged pointwise kernel (csrc/conv_pwdma_f32.hip): same sums
    for bm, bn in tiles:
        if bn == 129:                              # (the 256x128 tile with four 128x64 waves: plain form only -- no k-slices / tail split)
            out.append(((bm << 16) | bn, 16, -1, 0, 0))
            continue
        nt = -(-m // bm) * -(-d.cout // bn)
        stages = [16, 32] if (d.precision != _lib.PREC_F32 or (bm, bn) in ((256, 128), (128, 128), (128, 64), (64, 128))) else [16]
        splits = [-1] + ([2, 4] if (nt < 512 and d.k_pad >= 512) else []) + ([8] if (nt < 128 and d.k_pad >= 2048) else [])
        for st in stages:
            for sk in splits:
                out.append(((bm << 16) | bn, st, sk, 0, 0))
            # tail split (fp32 kernel): the main launch keeps whole rounds of workgroups (a round = 256 CUs x the 1..3
            # workgroups of this tile that fit a CU) and the leftover rows of tiles are cut along K so that they, too, occupy
            # the whole chip instead of costing one more, mostly empty round
            mt, ntc = -(-m // bm), -(-d.cout // bn)
            if d.precision == _lib.PREC_F32 and nt > 256 and d.k_pad >= 256:
                for round_tiles in (256, 512, 768):
                    main_rows = (nt // round_tiles) * round_tiles // ntc
                    tail_rows = mt - main_rows
                    tail_tiles = tail_rows * ntc
                    if main_rows <= 0 or not (0 < tail_tiles <= 0.6 * round_tiles):
                        continue
                    for target in (round_tiles, 2 * round_tiles):
                        ts = min(max(2, round(target / tail_tiles)), 16, d.k_pad // 64)
                        cand = ((bm << 16) | bn, st, -1, tail_rows, ts)
                        if ts > 1 and cand not in out:
                            out.append(cand)
    return out


TUNE_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune_gfx950.json")     # the shipped table
_TUNE_STATE = {"loaded": False, "table_id": None, "table_entries": 0, "table_path": None, "heuristic": 0, "explored": 0}


def _norm_entry(k, v):
    return tuple(k), tuple(v) + (0,) * (5 - len(v))


def load_tune_cache(path: str) -> int:
    """Merge a saved table of tuned configurations (signature -> choice) into this process."""
    import json
    with open(path) as f:
        _TUNE_CACHE.update(dict(_norm_entry(k, v) for k, v in json.load(f)))
    return len(_TUNE_CACHE)


def save_tune_cache(path: str) -> None:
    """Atomic (temp file + rename): several processes may read the table while one replaces it.  Entries are sorted, so the
    same choices always give the same file."""
    import json
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        json.dump(sorted([list(k), list(v)] for k, v in _TUNE_CACHE.items()), f)
    os.replace(tmp, path)


def _load_default_table() -> None:
    """Once per process: the table that ships with the package (PREMVOS_TUNE_TABLE=<file> names another one, =0 none) and,
    on top of it, PREMVOS_TUNE_CACHE when that file exists (bench.py's rank-0 table, tools/profile_round.sh)."""
    if _TUNE_STATE["loaded"]:
        return
    import hashlib
    _TUNE_STATE["loaded"] = True
    path = os.environ.get("PREMVOS_TUNE_TABLE", TUNE_TABLE)
    if path not in ("0", "", "none") and os.path.exists(path):
        with open(path, "rb") as f:
            raw = f.read()
        n0 = len(_TUNE_CACHE)
        load_tune_cache(path)
        _TUNE_STATE.update(table_id=hashlib.sha256(raw).hexdigest()[:16], table_entries=len(_TUNE_CACHE) - n0,
                           table_path=os.path.relpath(path, os.path.dirname(os.path.dirname(TUNE_TABLE))))
    cache_file = os.environ.get("PREMVOS_TUNE_CACHE")
    if cache_file and os.path.exists(cache_file):
        load_tune_cache(cache_file)


def tune_info() -> dict:
    """What decided the conv configurations of this process so far: the shipped table (sha-256 prefix of the file), how many
    signatures it did not hold and were configured by the closed-form rule (+ order-neutral timing), and how many were fully
    explored by wall clock (PREMVOS_AUTOTUNE=full only -- the one mode whose results may differ from run to run).  The
    drivers write this next to their outputs; bench.py puts it into its JSON line."""
    return {"table": _TUNE_STATE["table_path"], "table_sha256_16": _TUNE_STATE["table_id"],
            "table_entries": _TUNE_STATE["table_entries"], "signatures_by_rule": _TUNE_STATE["heuristic"],
            "signatures_explored_by_time": _TUNE_STATE["explored"],
            "mode": os.environ.get("PREMVOS_AUTOTUNE", "1")}


def numerics_key(d: ConvDesc, cand):
    """Two configurations of one layer with the same key add the same products in the same order (bit-identical outputs;
    tests/test_gpu_conv.py::test_order_neutral_knobs_are_bit_identical): within a kernel family the block / tile / stage depth only decide
    WHO computes an output element, never the order of its k-sum.  What does change the order: the family (implicit GEMM,
    small-N direct, Winograd F(2x2) slab / slab-free, F(4x4)), the number of k-slices, and which rows a tail split covers."""
    hint, st, sk, tail_rows, ts = cand
    if hint in (1, 2, 3, 4):
        return (hint,)
    if hint in (5, 6):                           # the streaming / LDS-DMA pointwise kernels add the products in the implicit GEMM's order
        return (0, None, None)
    bm = hint >> 16
    m = d.n * d.ho * d.wo
    st = st or 16
    kt = -(-d.k_pad // st)

    def slices(n):                               # the library cuts the KT stages into ceil(KT / ceil(KT / n)) slices of whole stages
        per = -(-kt // n)
        return (per * st, -(-kt // per))
    tail = (m - (-(-m // bm) - tail_rows) * bm,) + slices(ts) if (tail_rows > 0 and ts > 1) else None
    return (0, slices(sk) if sk > 1 else None, tail)


def rule_choice(d: ConvDesc):
    """The configuration of a layer signature that no table holds, as a closed-form function of the signature (so that every
    process, rank and run computes a frame with the same arithmetic): kernel family by shape, k-slices by tile count; the
    rules are the regularities of the measured tables (profiles/r0*_tune_choices.json)."""
    m = d.n * d.ho * d.wo
    cands = _candidates(d)
    fams = {c[0] for c in cands if c[0] in (1, 2, 3, 4)}
    if 1 in fams:
        return (1, 0, -1, 0, 0)                                  # 1-2 channel heads: the direct kernel
    if 4 in fams and m >= 16384:
        return (4, (0 if m >= 65536 else 64) + (16 if d.cin_pad < 512 else 0), -1, 0, 0)      # K-rich 3x3 (packed with F(4x4) filters): F(4x4,3x3); 16-deep stages for short K
    if 2 in fams and m < 8192:
        return (2, 64 if d.cout > 32 else 0, -1, 0, 0)           # coarse pyramid levels: too few blocks for the slab-free kernel
    if 3 in fams and d.cout > 32:
        blocks = [c[1] for c in cands if c[0] == 3]
        return (3, blocks[0], -1, 0, 0)                          # other 3x3 stride-1 (and atrous) layers: slab-free F(2x2,3x3)
    if 2 in fams and d.cout <= 32 and d.cin >= 32:
        return (2, 0, -1, 0, 0)
    bn = 32 if d.cout <= 32 else 64 if d.cout <= 64 else 128
    bm = 128 if -(-m // 128) * -(-d.cout // bn) >= 512 else 64
    if (bm, bn) == (64, 128) and -(-m // 64) * -(-d.cout // 128) < 128:
        bm, bn = 64, 64
    tiles = -(-m // bm) * -(-d.cout // bn)
    sk = -1
    if tiles < 384 and d.k_pad >= 512 and d.precision == _lib.PREC_F32:
        sk = 2 if tiles >= 192 else 4 if (tiles >= 64 or d.k_pad < 2048) else 8
    return ((bm << 16) | bn, 16, sk, 0, 0)


def _entry_for(d: ConvDesc):
    """The table / cache entry of this descriptor's signature -- unless THIS descriptor cannot run it.  ``_sig`` leaves out what
    only some kernels care about (pointer alignment of a channel window, the residual's pixel stride, the 2^30-element limit of the
    slab-free Winograd kernel): a layer with the signature of a tabled one but e.g. an unaligned slice falls back to the implicit
    GEMM -- for the streaming kernel (hint 5) with the SAME sums (same numerics_key), otherwise to the closed-form rule with the
    Winograd candidates withheld -- instead of failing in the library's argument check at launch time."""
    cand = _TUNE_CACHE[_sig(d)]
    hint = cand[0]
    # (PREMVOS_PWDMA=0 is the documented A/B switch of the LDS-DMA pointwise kernel: it must also silence the shipped table's
    #  hint-6 entries, not only the candidate list -- same sums either way)
    if (hint == 5 and not stream_applicable(d)) or \
            (hint == 6 and (not pwdma_applicable(d) or os.environ.get("PREMVOS_PWDMA", "1") == "0")):
        return ((128 << 16) | 128, 16, -1, 0, 0)
    if hint in (2, 3, 4) and cand not in _candidates(d):
        w2, w4, d.wgt_wino, d.wgt_wino4 = d.wgt_wino, d.wgt_wino4, None, None
        try:
            return rule_choice(d)
        finally:
            d.wgt_wino, d.wgt_wino4 = w2, w4
    return cand


def _out_digest(d: ConvDesc, lib, stream) -> int:
    """64-bit digest of the output window a launch of ``d`` wrote (premvos_digest_u64), on the host."""
    buf = torch.zeros(1, dtype=torch.int64, device=_lib.resolve_device())      # (autotune runs on the plan builder's thread and device)
    w = d.cout if d.out_mode == OUT_NHWC else d.cout_ps
    px = d.n * d.ho * d.wo * (1 if d.out_mode == OUT_NHWC else 4)
    _lib.check(lib.premvos_digest_u64(d.out, px, w, d.out_ps, buf.data_ptr(), stream), "digest")
    return int(buf.item())


def _time_cands(d: ConvDesc, cands, lib, stream, reps):
    """The fastest of ``cands`` by wall clock -- among candidates whose output is what their numerics key promises: a candidate that
    shares its key with an earlier one but writes different bits (a kernel bug, an uncovered shape) is dropped, loudly (ADVICE r03:
    timing alone once offered a configuration whose grid left rows unwritten)."""
    best, best_t = None, float("inf")
    ref_digest: dict = {}
    for cand in cands:
        d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = cand
        if lib.premvos_conv2d_f32(C.byref(d), stream) != 0:
            continue
        key, dig = numerics_key(d, cand), _out_digest(d, lib, stream)
        if ref_digest.setdefault(key, dig) != dig:
            import warnings
            warnings.warn(f"conv configuration {cand} of signature {_sig(d)} does not reproduce the output of its numerics class: dropped")
            continue
        t = float("inf")
        for _ in range(2):              # best of two bursts: a clock / scheduling hiccup must not pick the config
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                lib.premvos_conv2d_f32(C.byref(d), stream)
            b.record()
            b.synchronize()
            t = min(t, a.elapsed_time(b))
        if cand[0] == 2:
            t *= 1.08       # the slab Winograd moves 2-4x the HBM bytes of the slab-free one: it has to win clearly
        if t < best_t:
            best, best_t = cand, t
    return best


_POLISHED = set()


def _polish(descs, device, lib, stream):
    """PREMVOS_AUTOTUNE=polish (tools/make_tune_table.py --polish): re-time, with many repetitions, ONLY the order-neutral knobs
    (tile, stage depth, Winograd block) of signatures the table already holds -- candidates with the entry's own numerics_key --
    and replace an entry when another one is at least 1.5 % faster.  No result changes; used after a kernel change moves the
    balance between e.g. 16- and 32-deep stages."""
    todo = [d for d in descs if _sig(d) in _TUNE_CACHE and _sig(d) not in _POLISHED]
    if not todo:
        return
    need = 0
    for d in todo:
        for cand in _candidates(d):
            d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = cand
            need = max(need, workspace_bytes(d))
    ws = torch.empty(max(need // 4 + 1, 1), dtype=torch.float32, device=device)
    for d in todo:
        sig = _sig(d)
        if sig in _POLISHED:
            continue
        _POLISHED.add(sig)
        cur = _TUNE_CACHE[sig]
        if cur == (0, 0, 0, 0, 0) or cur[0] == 1:
            continue
        key = numerics_key(d, cur)
        same = [c for c in _candidates(d) if numerics_key(d, c) == key and c != cur]
        if not same:
            continue
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        times = {}
        for cand in [cur] + same:
            d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = cand
            if lib.premvos_conv2d_f32(C.byref(d), stream) != 0:
                continue
            t = float("inf")
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(12):
                    lib.premvos_conv2d_f32(C.byref(d), stream)
                b.record()
                b.synchronize()
                t = min(t, a.elapsed_time(b))
            times[cand] = t
        d.workspace, d.workspace_bytes = None, 0
        if cur in times:
            best = min(times, key=times.get)
            if best != cur and times[best] < 0.985 * times[cur]:
                _TUNE_CACHE[sig] = best
                _TUNE_STATE["explored"] += 1


def autotune(descs, device=None, reps: int = 4):
    """Freeze a (kernel family, tile, stage depth, k-split, tail split) configuration into every descriptor.

    Where the choice comes from, in this order (PREMVOS_AUTOTUNE=1, the default):
      1. the table that ships with the package (``tune_gfx950.json``: measured once per round on MI355X by
         tools/make_tune_table.py for the shapes of bench.py, the stage drivers and the streaming driver) -- no launch is timed;
      2. for a signature the table does not hold: ``rule_choice`` decides everything that changes the ORDER of the fp32 sums
         (family, k-slices), and only the order-neutral knobs (tile, stage depth, Winograd block) are timed on the real
         buffers.  Either way the arithmetic of a layer is a function of its signature and the table file alone, so two
         processes / ranks / runs write the same bytes (tests/test_gpu_plumbing.py::test_two_fresh_processes_write_identical_bytes).
    PREMVOS_AUTOTUNE=full explores every candidate by wall clock (cuDNN-find style; how the table is made: ~10 % faster plans
    than closed-form rules, but two runs may freeze kernels that differ in rounding); =0 leaves everything to the library's
    closed-form heuristics (no table, no timing)."""
    mode = os.environ.get("PREMVOS_AUTOTUNE", "1")
    if mode == "0" or not torch.cuda.is_available():
        return
    device = _lib.resolve_device(device)
    reps = int(os.environ.get("PREMVOS_AUTOTUNE_REPS", reps))      # launches per timing burst (tools/make_tune_table.py --reps)
    force = os.environ.get("PREMVOS_FORCE_KERNEL")          # diagnostics (tests/test_gpu_error_budget.py): every layer that CAN
    if force:                                                # run on this family does, whatever the table says
        fam = {"igemm": 0, "direct": 1, "wino": 2, "wino_fused": 3, "wino4": 4}[force]
        rest = []
        for d in descs:
            cands = [c for c in _candidates(d) if (c[0] if c[0] < 16 else 0) == fam and c[2] <= 0 and c[3] == 0]
            if cands and fam != 0:
                d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = cands[0]
            else:
                rest.append(d)
        descs = rest
        if fam == 0:                                         # "igemm": the closed-form implicit-GEMM choice, Winograd never
            for d in descs:
                c = rule_choice(d)
                if c[0] in (2, 3, 4):
                    w2, w4, d.wgt_wino, d.wgt_wino4 = d.wgt_wino, d.wgt_wino4, None, None
                    c = rule_choice(d)
                    d.wgt_wino, d.wgt_wino4 = w2, w4
                d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = c
            return
    with BUILD_LOCK:
        _load_default_table()
        lib = _lib.load()
        stream = _lib.current_stream()
        cache_file = os.environ.get("PREMVOS_TUNE_CACHE")
        if mode == "polish":
            _polish(descs, device, lib, stream)
        todo = [d for d in descs if _sig(d) not in _TUNE_CACHE]
        if todo and os.environ.get("PREMVOS_AUTOTUNE_FROZEN") == "1":
            # ranks > 0 of a multi-GPU job run rank 0's choices (same speed on every rank; the bits would agree anyway)
            raise _lib.PremvosError(f"{len(todo)} conv signature(s) are missing from the shared tune table {cache_file}")
        if todo:
            need = 0
            for d in todo:
                for cand in _candidates(d) + [rule_choice(d)]:
                    d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = cand
                    need = max(need, workspace_bytes(d))
            ws = torch.empty(max(need // 4 + 1, 1), dtype=torch.float32, device=device)
            for d in todo:
                sig = _sig(d)
                if sig in _TUNE_CACHE:
                    continue
                d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
                if mode == "full":
                    best = _time_cands(d, _candidates(d), lib, stream, reps)
                    _TUNE_STATE["explored"] += 1
                else:
                    rule = rule_choice(d)
                    key = numerics_key(d, rule)
                    same = [c for c in _candidates(d) if numerics_key(d, c) == key]
                    best = _time_cands(d, same, lib, stream, reps) if len(same) > 1 else None
                    best = best or rule
                    d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = best
                    if lib.premvos_conv2d_f32(C.byref(d), stream) != 0:      # a configuration the kernel refuses (a limit the rule
                        best = (0, 0, 0, 0, 0)                                # does not know): the library's own closed-form choice
                    _TUNE_STATE["heuristic"] += 1
                _TUNE_CACHE[sig] = best or (0, 0, 0, 0, 0)
                d.workspace, d.workspace_bytes = None, 0
            if cache_file and os.environ.get("RANK", "0") == "0":
                save_tune_cache(cache_file)
        for d in descs:
            d.tile_hint, d.stage_k, d.split_k, d.tail_m_tiles, d.tail_split_k = _entry_for(d)


@dataclass
class PackedConvS8:
    """Weights of one conv for csrc/conv_bf16x3_s8.hip: bf16 [cout_pad][kh*kw][ceil(cin/32)][4 groups][hi 8 | lo 8]."""
    wgt: torch.Tensor
    bias: Optional[torch.Tensor]
    cin: int
    cout: int
    kh: int
    kw: int
    cout_pad: int


def pack_conv_s8(weight: torch.Tensor, bias: Optional[torch.Tensor], device=None, scale: Optional[torch.Tensor] = None) -> PackedConvS8:
    """OIHW fp32 -> the S8 operand layout: row n = output channel, then (tap, block of 32 input channels, group of 8, hi | lo), zero
    padded; w = hi + lo with hi = bf16(w), lo = bf16(w - hi).  ``scale`` folds a frozen BatchNorm's gamma / sqrt(var + eps) in."""
    device = _lib.resolve_device(device)
    w = weight.detach().to(torch.float32).cpu()
    cout, cin, kh, kw = w.shape
    if scale is not None:
        w = w * scale.detach().to(torch.float32).cpu().view(-1, 1, 1, 1)
    cout_pad, kc = _r(cout, 256), -(-cin // 32)          # (whole 256-row tiles: the ping-pong kernel addresses weight rows by a fixed stride)
    full = torch.zeros((cout_pad, kh * kw, kc * 32), dtype=torch.float32)
    full[:cout, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    hi = full.to(torch.bfloat16)
    lo = (full - hi.float()).to(torch.bfloat16)
    both = torch.stack([hi.view(cout_pad, kh * kw, kc, 4, 8), lo.view(cout_pad, kh * kw, kc, 4, 8)], dim=4)     # [..., group, part, 8]
    b = None
    if bias is not None:
        b = torch.zeros(cout_pad, dtype=torch.float32)
        b[:cout] = bias.detach().to(torch.float32).cpu()
        b = b.to(device)
    return PackedConvS8(both.contiguous().to(device), b, cin, cout, kh, kw, cout_pad)


def conv_s8_desc(x: NHWC, pk: PackedConvS8, out: Optional[NHWC], out_s8: Optional[NHWC], stride=(1, 1), dilation=(1, 1), pad=(0, 0),
                 act=ACT_NONE, slope=0.1, res: Optional[NHWC] = None) -> ConvDesc:
    """Geometry / epilogue descriptor of one S8 conv launch (premvos_conv_bf16x3_s8_f32); the output size comes from the outputs."""
    o = out if out is not None else out_s8
    # (an S8 window may carry up to 7 zero channels behind the layer's own: the packed weights are zero there)
    assert x.layout == "s8" and x.c == _r(pk.cin, 8) and x.coff % 8 == 0, (x.layout, x.c, pk.cin, x.coff)
    assert o is not None and o.c == pk.cout and o.n == x.n and pk.cout % 8 == 0
    assert out is None or out.layout == "f32"
    assert out_s8 is None or (out_s8.layout == "s8" and (out_s8.n, out_s8.h, out_s8.w, out_s8.c) == (o.n, o.h, o.w, o.c) and out_s8.coff % 8 == 0)
    assert res is None or (res.layout == "f32" and (res.n, res.h, res.w, res.c) == (o.n, o.h, o.w, o.c))
    d = ConvDesc()
    d.inp, d.wgt = None, None
    d.bias = pk.bias.data_ptr() if pk.bias is not None else None
    d.res = res.ptr if res is not None else None
    d.out = out.ptr if out is not None else None
    d.n, d.h, d.w, d.cin, d.in_ps = x.n, x.h, x.w, x.c, x.ps
    d.ho, d.wo, d.cout = o.h, o.w, pk.cout
    d.out_ps = out.ps if out is not None else 0
    d.res_ps = res.ps if res is not None else 0
    d.kh, d.kw = pk.kh, pk.kw
    d.sh, d.sw = stride
    d.dh, d.dw = dilation
    d.pt, d.pl = pad
    d.cin_pad, d.k_pad, d.cout_pad = _r(pk.cin, 32), pk.kh * pk.kw * _r(pk.cin, 32), pk.cout_pad
    d.act, d.slope, d.out_mode, d.cout_ps = act, slope, OUT_NHWC, 0
    d.precision = _lib.PREC_BF16X3
    return d


S8_HINT = 6          # ConvDesc.tile_hint of a launch that goes to premvos_conv_bf16x3_s8_f32 (a marker for tables / reports: that entry has one kernel)


def s8_tile_rule(m: int, cout: int) -> int:
    """Tile of an S8 conv as a closed-form function of its shape (no timing: the same kernel on every rank and run; every tile adds
    the products in the same order, so the choice never changes a result).  From tools/dev/s8_bench.py on MI355X
    (profiles/r04_s8_bench.txt): 256 x 256 / eight waves wherever the layer is wider than one 128-column tile; 128 x 128 with two
    workgroups per CU (64 KB of LDS each: one's epilogue under the other's K loop) for cout <= 128."""
    forced = os.environ.get("PREMVOS_S8_TILE")               # developer A/B runs only
    if forced:
        return int(forced)
    return 0 if cout > 128 else 5


def conv_s8(x: NHWC, pk: PackedConvS8, out: Optional[NHWC] = None, out_s8: Optional[NHWC] = None, tile: Optional[int] = None,
            res_s8: Optional[NHWC] = None, **kw):
    d = conv_s8_desc(x, pk, out, out_s8, **kw)
    run_s8(d, x, pk, out_s8, tile, res_s8=res_s8)
    return out if out is not None else out_s8


def run_s8(d: ConvDesc, x: NHWC, pk: PackedConvS8, out_s8: Optional[NHWC], tile: Optional[int] = None, stream: Optional[int] = None,
           res_s8: Optional[NHWC] = None):
    """``res_s8``: the residual as an S8 tensor (instead of ``res`` of the descriptor): added as hi + lo."""
    t = s8_tile_rule(d.n * d.ho * d.wo, d.cout) if tile is None else tile
    assert res_s8 is None or (res_s8.layout == "s8" and res_s8.c == d.cout and res_s8.coff % 8 == 0 and not d.res)
    _lib.check(_lib.load().premvos_conv_bf16x3_s8_f32(C.byref(d), x.ptr, pk.wgt.data_ptr(), out_s8.ptr if out_s8 is not None else None,
                                                      out_s8.ps if out_s8 is not None else 0,
                                                      res_s8.ptr if res_s8 is not None else None, res_s8.ps if res_s8 is not None else 0, t,
                                                      _lib.current_stream() if stream is None else stream), "conv_bf16x3_s8")


def split8(x: NHWC, out: NHWC):
    """fp32 NHWC -> S8 (same shape): the entry of an S8 chain whose producer is an fp32 kernel."""
    assert x.layout == "f32" and out.layout == "s8" and (x.n, x.h, x.w, _r(x.c, 8)) == (out.n, out.h, out.w, out.c) and out.coff % 8 == 0
    _lib.check(_lib.load().premvos_split8_f32(x.ptr, x.ps, out.ptr, out.ps, x.n * x.h * x.w, x.c, _lib.current_stream()), "split8")
    return out


def conv2d(x: NHWC, pk: PackedConv, out: NHWC, **kw):
    d = conv_desc(x, pk, out, **kw)
    ws = assign_workspace([d], x.buf.device)          # noqa: F841  (kept alive until the launch is enqueued)
    _lib.check(_lib.load().premvos_conv2d_f32(C.byref(d), _lib.current_stream()), "conv2d")
    return out


def run_desc(d: ConvDesc, stream: Optional[int] = None):
    _lib.check(_lib.load().premvos_conv2d_f32(C.byref(d), _lib.current_stream() if stream is None else stream),
               "conv2d")


def vslab_enabled() -> bool:
    """PREMVOS_VSLAB=0: every F(4x4) layer of a DenseNet block transforms its whole input again (rounds 2-3)."""
    return os.environ.get("PREMVOS_VSLAB", "1") != "0"


def wino4_slab_plan(layers):
    """Kept input-transform slab for the Winograd F(4x4,3x3) layers of ONE DenseNet concat buffer (PWCNet.py:201-264: layer i
    reads everything layers 0 ... i-1 produced; include/premvos_hip.h, premvos_conv_wino4_slab_f32).  ``layers``: (desc, first
    channel of the layer's input window in the buffer) in launch order, every desc with tile_hint 4; the windows must start at
    multiples of 16 and end at the same channel (then the zero K-padding the first layer writes serves all).  Returns
    (floats of slab needed, [(desc, v_pitch, v_c0, t_cn)]) or None when the block does not have that shape."""
    if not layers:
        return None
    ends = {off + d.cin for d, off in layers}
    tiles = {d.n * ((d.ho + 3) // 4) * ((d.wo + 3) // 4) for d, _ in layers}
    if len(ends) != 1 or len(tiles) != 1 or any(off % 16 or d.tile_hint != 4 or (d.dh, d.dw) != (1, 1) for d, off in layers):
        return None
    kp = lambda d: _r(d.cin_pad, 16)
    pitch = max(off + kp(d) for d, off in layers)
    plan, lo = [], None
    for d, off in layers:
        t_cn = kp(d) if lo is None else max(0, lo - off)          # first layer: the whole window incl. the zero K padding
        lo = off if lo is None else min(lo, off)
        plan.append((d, pitch, off, t_cn))
    return 36 * tiles.pop() * pitch, plan


def run_wino4_slab(d: ConvDesc, slab: torch.Tensor, v_pitch: int, v_c0: int, t_cn: int, stream: Optional[int] = None):
    _lib.check(_lib.load().premvos_conv_wino4_slab_f32(C.byref(d), slab.data_ptr(), slab.numel() * 4, v_pitch, v_c0, t_cn,
                                                       _lib.current_stream() if stream is None else stream), "conv_wino4_slab")


def corr(f1: NHWC, f2: NHWC, out: NHWC, md: int = 4, slope: float = 1.0, copy_f1: bool = False):
    assert (f1.n, f1.h, f1.w, f1.c) == (f2.n, f2.h, f2.w, f2.c)
    _lib.check(_lib.load().premvos_corr_fwd_f32(f1.ptr, f1.ps, f2.ptr, f2.ps, out.ptr, out.ps, f1.n, f1.h, f1.w,
                                                f1.c, md, slope, int(copy_f1), _lib.current_stream()), "corr")
    return out


def warp_corr(f1: NHWC, x2: NHWC, flow: NHWC, scale: float, out: NHWC, md: int = 4, slope: float = 1.0,
              copy_f1: bool = False):
    """corr(f1, warp(x2, flow * scale)) in one kernel (bit-identical to the two calls)."""
    assert (f1.n, f1.h, f1.w, f1.c) == (x2.n, x2.h, x2.w, x2.c) and flow.c == 2
    assert (f1.n, f1.h, f1.w) == (flow.n, flow.h, flow.w)
    _lib.check(_lib.load().premvos_warp_corr_fwd_f32(f1.ptr, f1.ps, x2.ptr, x2.ps, flow.ptr, flow.ps, scale, out.ptr,
                                                     out.ps, f1.n, f1.h, f1.w, f1.c, md, slope, int(copy_f1),
                                                     _lib.current_stream()), "warp_corr")
    return out


def corr_nchw(in1: torch.Tensor, in2: torch.Tensor, pad_size: int, kernel_size: int, max_displacement: int,
              stride1: int, stride2: int, corr_multiply: int = 1) -> torch.Tensor:
    """Op-level twin of ``corr_cuda_forward`` on NCHW tensors (allocates the output itself, like
    the reference which resizes the caller's empty tensor, corr_cuda.c:52)."""
    import math
    n, c, h, w = in1.shape
    kr = (kernel_size - 1) // 2
    border = max_displacement + kr
    ow = int(math.ceil((w + 2 * pad_size - 2 * border) / float(stride1)))
    oh = int(math.ceil((h + 2 * pad_size - 2 * border) / float(stride1)))
    dd = 2 * (max_displacement // stride2) + 1
    out = torch.empty((n, dd * dd, oh, ow), dtype=torch.float32, device=in1.device)
    _lib.check(_lib.load().premvos_corr_nchw_fwd_f32(
        in1.contiguous().data_ptr(), in2.contiguous().data_ptr(), out.data_ptr(), n, c, h, w, pad_size,
        kernel_size, max_displacement, stride1, stride2, corr_multiply, _lib.current_stream()), "corr_nchw")
    return out


def warp(x: NHWC, flow: NHWC, scale: float, out: NHWC):
    assert flow.c == 2 and (x.n, x.h, x.w) == (flow.n, flow.h, flow.w)
    _lib.check(_lib.load().premvos_warp_fwd_f32(x.ptr, x.ps, flow.ptr, flow.ps, scale, out.ptr, out.ps, x.n, x.h,
                                                x.w, x.c, _lib.current_stream()), "warp")
    return out


def nchw_to_nhwc(src: torch.Tensor, out: NHWC):
    n, c, h, w = src.shape
    assert src.is_contiguous() and (out.n, out.h, out.w) == (n, h, w) and out.c >= c and out.coff == 0
    _lib.check(_lib.load().premvos_nchw_to_nhwc_f32(src.data_ptr(), out.ptr, out.ps, n, c, h, w,
                                                    _lib.current_stream()), "nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: NHWC, dst: Optional[torch.Tensor] = None) -> torch.Tensor:
    if dst is None:
        dst = torch.empty((x.n, x.c, x.h, x.w), dtype=torch.float32, device=x.buf.device)
    _lib.check(_lib.load().premvos_nhwc_to_nchw_f32(x.ptr, x.ps, dst.data_ptr(), x.n, x.c, x.h, x.w,
                                                    _lib.current_stream()), "nhwc_to_nchw")
    return dst


def out_size(size: int, k: int, stride: int, pad_lo: int, pad_hi: int, dil: int = 1) -> int:
    return (size + pad_lo + pad_hi - dil * (k - 1) - 1) // stride + 1


def hw_tuple(v) -> Tuple[int, int]:
    return (v, v) if isinstance(v, int) else tuple(v)
