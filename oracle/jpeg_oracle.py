"""TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench's cpu_baseline leg may import this; the product never does).

CPU restatement of the baseline JPEG decode the reference's stages run through their image readers: ``cv2.imread``
(proposal_net/train.py:500), ``scipy.ndimage.imread`` = PIL (optical_flow_net-PWC-Net/script_pwc_multi.py:34) and PIL
(ReID_net/prepare_input.py:38) -- all of them libjpeg / libjpeg-turbo with its defaults (JDCT_ISLOW, fancy up-sampling,
YCbCr -> RGB).  The decoder itself is a third-party dependency that is not part of /root/reference (libjpeg-turbo, the copy
inside this image's Pillow 12.2 reports libjpeg 6.2 API); this file restates its published algorithm:

  * ITU-T T.81 baseline sequential Huffman decoding (sections F.2.2 / F.2.4.4: DC differences, AC run/size, EOB, ZRL,
    restart intervals),
  * jidctint.c ``jpeg_idct_islow`` (Loeffler-Ligtenberg-Moshovitz, CONST_BITS 13, PASS1_BITS 2) behind the de-quantisation,
    with the saturating output of the SIMD build (clamp(x, -128, 127) + 128),
  * jdsample.c ``h2v1_fancy_upsample`` / ``h2v2_fancy_upsample`` (triangle filter; first / last column special cases; the row
    above the first / below the last real row is that row again, jdmainct.c),
  * jdcolor.c ``ycc_rgb_convert`` (16-bit fixed point tables).

PARITY: pinned against the library itself -- tests/test_cpu_jpeg.py decodes files with PIL (the very libjpeg-turbo the default
product path uses) and requires ``decode`` to return the same bytes, for 4:4:4 / 4:2:2 / 4:2:0 / grey files, odd sizes, custom
Huffman tables and restart markers.  Progressive, arithmetic-coded, 12-bit, CMYK and multi-scan files are refused
(``Unsupported``): the product's optional GPU decoder falls back to the default reader for them.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                   28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                   54, 47, 55, 62, 63])


class Unsupported(ValueError):
    pass


def parse(data: bytes) -> Dict[str, object]:
    """Marker segments up to and including SOS (T.81 annex B).  Returns frame geometry, tables and the entropy-coded bytes."""
    if data[:2] != b"\xff\xd8":
        raise ValueError("not a JPEG (no SOI)")
    pos, qt, huff, frame, dri, adobe = 2, {}, {}, None, 0, None
    while True:
        while data[pos] != 0xFF:
            pos += 1
        while data[pos] == 0xFF:
            pos += 1
        m = data[pos]
        pos += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        if m == 0xD9:
            raise ValueError("EOI before SOS")
        (ln,) = struct.unpack_from(">H", data, pos)
        seg = data[pos + 2:pos + ln]
        pos += ln
        if m == 0xDB:                                              # DQT
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                if pq:
                    vals = struct.unpack_from(">64H", seg, i)
                    i += 128
                else:
                    vals = seg[i:i + 64]
                    i += 64
                t = np.zeros(64, np.int64)
                t[ZIGZAG] = np.array(list(vals), np.int64)         # stored in zig-zag order; kept in natural order
                qt[tq] = t
        elif m == 0xC4:                                            # DHT
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                counts = list(seg[i + 1:i + 17])
                n = sum(counts)
                syms = list(seg[i + 17:i + 17 + n])
                i += 17 + n
                table, code, k = {}, 0, 0
                for ln_ in range(1, 17):                           # T.81 annex C: canonical codes in order of length
                    for _ in range(counts[ln_ - 1]):
                        table[(ln_, code)] = syms[k]
                        code += 1
                        k += 1
                    code <<= 1
                huff[(tc, th)] = table
        elif m in (0xC0, 0xC1):                                    # SOF0 / SOF1: sequential Huffman
            p, h, w, nc = struct.unpack_from(">BHHB", seg, 0)
            if p != 8:
                raise Unsupported(f"{p}-bit samples")
            comps = [(seg[6 + 3 * c], seg[7 + 3 * c] >> 4, seg[7 + 3 * c] & 15, seg[8 + 3 * c]) for c in range(nc)]
            frame = {"height": h, "width": w, "comps": comps}
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise Unsupported(f"SOF marker 0x{m:02x} (progressive / lossless / arithmetic)")
        elif m == 0xDD:
            (dri,) = struct.unpack_from(">H", seg, 0)
        elif m == 0xEE and seg[:5] == b"Adobe":
            adobe = seg[11]
        elif m == 0xDA:                                            # SOS
            if frame is None:
                raise ValueError("SOS before SOF")
            ns = seg[0]
            if ns != len(frame["comps"]):
                raise Unsupported("multi-scan file")
            sel = {seg[1 + 2 * k]: (seg[2 + 2 * k] >> 4, seg[2 + 2 * k] & 15) for k in range(ns)}
            if [seg[1 + 2 * k] for k in range(ns)] != [c[0] for c in frame["comps"]]:
                raise Unsupported("scan component order")
            break
    comps = frame["comps"]
    if len(comps) not in (1, 3):
        raise Unsupported(f"{len(comps)} components")
    if len(comps) == 3:
        if adobe is not None and adobe != 1:
            raise Unsupported("Adobe transform flag (RGB / CMYK data)")
        if adobe is None and [c[0] for c in comps] == [82, 71, 66]:
            raise Unsupported("RGB component ids")
        hs, vs = [c[1] for c in comps], [c[2] for c in comps]
        if (hs[1], vs[1], hs[2], vs[2]) != (1, 1, 1, 1) or (hs[0], vs[0]) not in ((1, 1), (2, 1), (2, 2)):
            raise Unsupported(f"sampling factors {hs} x {vs}")
    else:
        comps = [(comps[0][0], 1, 1, comps[0][3])]                 # a one-component scan is never interleaved: 1x1 MCUs
    frame["comps"] = comps
    frame.update(qt=qt, huff=huff, dri=dri, sel=sel, scan=data[pos:])
    return frame


class _Bits:
    def __init__(self, data: bytes):
        self.d, self.p, self.acc, self.n = data, 0, 0, 0

    def _fill(self):
        d = self.d
        b = d[self.p] if self.p < len(d) else 0xFF
        if b == 0xFF:
            nxt = d[self.p + 1] if self.p + 1 < len(d) else 0xD9
            if nxt == 0:
                self.p += 2
            else:
                b = 0                                              # a marker: feed zeros (the data should have ended)
        else:
            self.p += 1
        self.acc = (self.acc << 8) | b
        self.n += 8

    def bit(self) -> int:
        if self.n == 0:
            self._fill()
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k: int) -> int:
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v

    def restart(self):
        """Drop the padding bits and step over the RSTn marker."""
        self.acc = self.n = 0
        d = self.d
        while self.p < len(d) and not (d[self.p] == 0xFF and 0xD0 <= d[self.p + 1] <= 0xD7):
            self.p += 1
        self.p += 2


def _symbol(br: _Bits, table) -> int:
    code = 0
    for ln in range(1, 17):
        code = (code << 1) | br.bit()
        s = table.get((ln, code))
        if s is not None:
            return s
    raise ValueError("bad Huffman code")


def _extend(v: int, t: int) -> int:                                # T.81 F.2.2.1
    return v if v >= (1 << (t - 1)) else v - (1 << t) + 1


def entropy_decode(f: Dict[str, object]) -> List[np.ndarray]:
    """Quantised coefficients per component: int16 [blocks_h][blocks_w][64] in natural (row-major) order."""
    comps = f["comps"]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mcux = -(-f["width"] // (8 * hmax))
    mcuy = -(-f["height"] // (8 * vmax))
    coef = [np.zeros((mcuy * c[2], mcux * c[1], 64), np.int16) for c in comps]
    br, pred, dri = _Bits(f["scan"]), [0] * len(comps), f["dri"]
    for mcu in range(mcux * mcuy):
        if dri and mcu and mcu % dri == 0:
            br.restart()
            pred = [0] * len(comps)
        my, mx = divmod(mcu, mcux)
        for ci, (cid, h, v, tq) in enumerate(comps):
            dc_t, ac_t = f["huff"][(0, f["sel"][cid][0])], f["huff"][(1, f["sel"][cid][1])]
            for by in range(v):
                for bx in range(h):
                    blk = coef[ci][my * v + by, mx * h + bx]
                    t = _symbol(br, dc_t)
                    pred[ci] += _extend(br.bits(t), t) if t else 0
                    blk[0] = pred[ci]
                    k = 1
                    while k < 64:
                        rs = _symbol(br, ac_t)
                        r, s = rs >> 4, rs & 15
                        if s == 0:
                            if r != 15:
                                break                              # EOB
                            k += 16                                # ZRL
                            continue
                        k += r
                        blk[ZIGZAG[k]] = _extend(br.bits(s), s)
                        k += 1
    return coef


def _fix(x: float) -> int:
    return int(x * (1 << 13) + 0.5)


def _idct_1d(d: List[np.ndarray], shift: int, first: bool) -> List[np.ndarray]:
    """One pass of jidctint.c over eight int64 arrays (a column / a row of every block at once)."""
    z2, z3 = d[2], d[6]
    z1 = (z2 + z3) * _fix(0.541196100)
    tmp2 = z1 + z3 * -_fix(1.847759065)
    tmp3 = z1 + z2 * _fix(0.765366865)
    tmp0, tmp1 = (d[0] + d[4]) << 13, (d[0] - d[4]) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = d[7], d[5], d[3], d[1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * _fix(1.175875602)
    tmp0, tmp1 = tmp0 * _fix(0.298631336), tmp1 * _fix(2.053119869)
    tmp2, tmp3 = tmp2 * _fix(3.072711026), tmp3 * _fix(1.501321110)
    z1, z2 = z1 * -_fix(0.899976223), z2 * -_fix(2.562915447)
    z3, z4 = z3 * -_fix(1.961570560) + z5, z4 * -_fix(0.390180644) + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    out = [tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3]
    return [(o + (1 << (shift - 1))) >> shift for o in out]


def idct_islow(coef: np.ndarray, quant: np.ndarray) -> np.ndarray:
    """int16 [bh][bw][64] quantised coefficients -> uint8 plane [bh*8][bw*8] (jidctint.c jpeg_idct_islow)."""
    bh, bw = coef.shape[:2]
    x = (coef.astype(np.int64) * quant).reshape(bh, bw, 8, 8)
    ws = _idct_1d([x[:, :, r, :] for r in range(8)], 13 - 2, True)            # pass 1: columns (index r = row of the input)
    ws = np.stack(ws, axis=2)                                                  # [bh][bw][row][col]
    out = _idct_1d([ws[:, :, :, c] for c in range(8)], 13 + 2 + 3, False)     # pass 2: rows
    out = np.stack(out, axis=3)
    out = np.clip(out, -128, 127) + 128
    return out.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8).astype(np.uint8)


def upsample_h2v1(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v1_fancy_upsample on a [rows][dw] plane -> [rows][2*dw]."""
    a = p.astype(np.int64)
    left = np.concatenate([a[:, :1], a[:, :-1]], axis=1)
    right = np.concatenate([a[:, 1:], a[:, -1:]], axis=1)
    even, odd = (3 * a + left + 1) >> 2, (3 * a + right + 2) >> 2
    even[:, 0], odd[:, -1] = a[:, 0], a[:, -1]
    return np.stack([even, odd], axis=2).reshape(a.shape[0], -1)


def upsample_h2v2(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v2_fancy_upsample on the REAL rows / columns of a chroma plane [dh][dw] -> [2*dh][2*dw]."""
    a = p.astype(np.int64)
    up = np.concatenate([a[:1], a[:-1]], axis=0)
    down = np.concatenate([a[1:], a[-1:]], axis=0)
    rows = np.stack([3 * a + up, 3 * a + down], axis=1).reshape(-1, a.shape[1])            # column sums of each output row
    left = np.concatenate([rows[:, :1], rows[:, :-1]], axis=1)
    right = np.concatenate([rows[:, 1:], rows[:, -1:]], axis=1)
    even, odd = (3 * rows + left + 8) >> 4, (3 * rows + right + 7) >> 4
    even[:, 0], odd[:, -1] = (4 * rows[:, 0] + 8) >> 4, (4 * rows[:, -1] + 7) >> 4
    return np.stack([even, odd], axis=2).reshape(rows.shape[0], -1)


def ycc_to_rgb(y: np.ndarray, cb: np.ndarray, cr: np.ndarray) -> np.ndarray:
    """jdcolor.c build_ycc_rgb_table + ycc_rgb_convert."""
    half, x = 1 << 15, np.arange(256, dtype=np.int64) - 128
    fix = lambda v: int(v * 65536 + 0.5)
    cr_r, cb_b = (fix(1.40200) * x + half) >> 16, (fix(1.77200) * x + half) >> 16
    cr_g, cb_g = -fix(0.71414) * x, -fix(0.34414) * x + half
    yy = y.astype(np.int64)
    r = yy + cr_r[cr]
    g = yy + ((cb_g[cb] + cr_g[cr]) >> 16)
    b = yy + cb_b[cb]
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def reconstruct(f: Dict[str, object], coef: List[np.ndarray]) -> np.ndarray:
    """Coefficients -> uint8 [H][W][3] RGB (a grey file is replicated, PIL's ``convert('RGB')``)."""
    h, w, comps = f["height"], f["width"], f["comps"]
    planes = [idct_islow(coef[i], f["qt"][c[3]]) for i, c in enumerate(comps)]
    if len(comps) == 1:
        g = planes[0][:h, :w]
        return np.stack([g, g, g], axis=-1)
    hs, vs = comps[0][1], comps[0][2]
    dw, dh = -(-w // hs), -(-h // vs)                              # downsampled_width / height of the chroma components
    if hs == 2 and dw <= 2:
        raise Unsupported("chroma plane narrower than 3 samples (libjpeg replicates instead of filtering)")
    ch = []
    for p in planes[1:]:
        p = p[:dh, :dw]
        if (hs, vs) == (2, 2):
            p = upsample_h2v2(p)
        elif (hs, vs) == (2, 1):
            p = upsample_h2v1(p)
        ch.append(p[:h, :w])
    return ycc_to_rgb(planes[0][:h, :w], ch[0], ch[1])


def decode(data: bytes) -> np.ndarray:
    f = parse(data)
    return reconstruct(f, entropy_decode(f))
