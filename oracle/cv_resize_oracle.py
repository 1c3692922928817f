"""CPU restatement of the two cv2.resize flavours on the flow stage boundary.  TEST INFRASTRUCTURE ONLY.

script_pwc_multi.py:45 resizes uint8 frames with ``cv2.resize(im, (W_, H_))`` and :63-64 resizes the
float32 flow planes with ``cv2.resize(flo, (W, H))`` (default INTER_LINEAR).  OpenCV is a third-party
dependency that is absent from /root/reference AND from this image (``import cv2`` fails), so the
algorithm is restated from OpenCV's published implementation (modules/imgproc/src/resize.cpp,
3.x/4.x ``resizeGeneric_`` + ``HResizeLinear`` / ``VResizeLinear``):

  * source coordinate  fx = (dx + 0.5) * (src/dst) - 0.5 (double, cast to float), sx = floor(fx);
    sx < 0 -> (sx, fx) = (0, 0);  sx >= src-1 -> (src-1, 0)
  * uint8: coefficients are 11-bit fixed point, cvRound((1-fx)*2048), cvRound(fx*2048) (round half
    to even); horizontal pass keeps ints; vertical pass
        dst = ( ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2 ) >> 2
  * float32: plain float lerp, horizontal then vertical.

PARITY UNPINNED: no cv2 and no reference fixture exists to check these two functions against; the
HIP kernels are tested bit-exactly (uint8) / to 1e-6 (float) against this restatement.
"""
from __future__ import annotations

import numpy as np


def _coeffs(dst: int, src: int):
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, f


def resize_linear_u8(img: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, _ = img.shape
    if (h, w) == (dst_h, dst_w):
        return img.copy()
    x0, x1, fx = _coeffs(dst_w, w)
    y0, y1, fy = _coeffs(dst_h, h)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int32)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int32)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int32)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int32)
    src = img.astype(np.int32)
    rows = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]      # [h, dst_w, c]
    r0, r1 = rows[y0], rows[y1]
    v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def resize_linear_f32(img: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    assert img.dtype == np.float32 and img.ndim == 2
    h, w = img.shape
    x0, x1, fx = _coeffs(dst_w, w)
    y0, y1, fy = _coeffs(dst_h, h)
    a0, a1 = (np.float32(1) - fx), fx
    b0, b1 = (np.float32(1) - fy), fy
    rows = img[:, x0] * a0[None, :] + img[:, x1] * a1[None, :]
    return (rows[y0] * b0[:, None] + rows[y1] * b1[:, None]).astype(np.float32)


def flow_preprocess(im1: np.ndarray, im2: np.ndarray):
    """script_pwc_multi.py:34-56 -> float32 [1,6,H_,W_] (numpy), plus (H_, W_)."""
    from math import ceil
    h, w = im1.shape[:2]
    h_, w_ = int(ceil(h / 64.0) * 64), int(ceil(w / 64.0) * 64)
    planes = []
    for im in (im1, im2):
        r = resize_linear_u8(np.ascontiguousarray(im[:, :, :3]), w_, h_)
        r = r[:, :, ::-1]
        r = (1.0 * r / 255.0)
        planes.append(np.transpose(r, (2, 0, 1)).astype(np.float32))
    return np.concatenate(planes, 0)[None], h_, w_


def flow_postprocess(flow2: np.ndarray, h: int, w: int, h_: int, w_: int) -> np.ndarray:
    """script_pwc_multi.py:59-68: flow2 [2,h4,w4] -> [h,w,2]."""
    flo = (flow2 * np.float32(20.0)).astype(np.float32)
    u = resize_linear_f32(np.ascontiguousarray(flo[0]), w, h)
    v = resize_linear_f32(np.ascontiguousarray(flo[1]), w, h)
    u *= np.float32(w / float(w_))
    v *= np.float32(h / float(h_))
    return np.dstack((u, v))
