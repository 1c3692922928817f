"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the mask operations MergeTrack performs on the hot path's
outputs.  Only tests/ may import this; the product (premvos_amd/mergetrack.py) calls HIP kernels and has no fallback.

Restated from:
  * MergeTrack/merge_functions.py:197-207  get_flow           (.flo reader)
  * MergeTrack/merge_functions.py:209-217  warp_flow          (cv2.remap INTER_LINEAR on a uint8 mask, then == 1)
  * MergeTrack/merge_functions.py:219-241  warp_proposals     (RLE / bbox / score bookkeeping of the warped masks)
  * MergeTrack/merge_functions.py:38-45    the warp-score term of calculate_scores (pycocotools iou, iscrowd = 0)

Third-party arithmetic that is NOT under /root/reference and is absent from this image (cv2, pycocotools), restated
from their published algorithms:
  * OpenCV remap, CV_8U + INTER_LINEAR + CV_32FC2 map: coordinates -> 1/32 pixel fixed point with cvRound (half to even),
    integer cell = arithmetic >> 5 saturated to int16, 5-bit fractions; bilinear tap weights in 15-bit fixed point
    ((32-a)|a products * 32, sum 32768); BORDER_CONSTANT 0; value = (sum w*v + 2^14) >> 15.
  * COCO maskApi.c rleIou (iscrowd = 0): i = |A and B|, u = |A or B|, u = 1 when i == 0, iou = i/u in double;
    rleEncode / rleToBbox as in premvos_amd/rle.py (checked there).
PARITY: the reference holds no test or golden vector for these functions and neither library can be imported here.  The
COMPOSITION is pinned since round 2 by executing merge_functions.py itself (tools/make_golden_merge.py: a recording cv2.remap
keeps the sampling map the reference builds, integer-valued flows make bilinear == gather; dense-mask stand-ins for pycocotools):
sign / grid of the warp, `== 1` binarisation, warp_proposals' keys and scores (tests/test_cpu_merge.py, HIP twin in
tests/test_gpu_merge.py).  OpenCV's 1/32-pixel fixed-point interpolation and rleIou stay PARITY UNPINNED: restated from the
published algorithms, analytic pins only (identity / integer-shift flows, quantisation and half-up cases, hand-computed IoUs).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np


def get_flow(filename: str) -> np.ndarray:
    with open(filename, "rb") as f:
        magic = np.frombuffer(f.read(4), np.float32)[0]
        assert magic == np.float32(202021.25), "Magic number incorrect. Invalid .flo file"
        w = int(np.frombuffer(f.read(4), np.int32)[0])
        h = int(np.frombuffer(f.read(4), np.int32)[0])
        return np.resize(np.frombuffer(f.read(8 * w * h), np.float32), (h, w, 2)).copy()


def remap_map(flow: np.ndarray) -> np.ndarray:
    """merge_functions.py:211-214: float32 map = -flow, += arange (numpy adds in float64, stores float32)."""
    h, w = flow.shape[:2]
    m = -flow.astype(np.float32)
    m[:, :, 0] += np.arange(w)
    m[:, :, 1] += np.arange(h)[:, np.newaxis]
    return m


def remap_linear_u8(img: np.ndarray, mp: np.ndarray) -> np.ndarray:
    """cv2.remap(img, mp, None, cv2.INTER_LINEAR) for a 2-D uint8 image and a float32 [h,w,2] map."""
    h, w = img.shape
    s = np.rint(mp.astype(np.float32) * np.float32(32)).astype(np.int64)          # rint = half to even = cvRound
    sx, sy = s[..., 0], s[..., 1]
    ix, iy = np.clip(sx >> 5, -32768, 32767), np.clip(sy >> 5, -32768, 32767)
    ax, ay = sx & 31, sy & 31
    pad = np.zeros((h + 2, w + 2), np.int64)
    pad[1:-1, 1:-1] = img

    def tap(yy, xx):
        ok = (yy >= -1) & (yy <= h) & (xx >= -1) & (xx <= w)
        return np.where(ok, pad[np.clip(yy + 1, 0, h + 1), np.clip(xx + 1, 0, w + 1)], 0)

    acc = ((32 - ay) * (32 - ax) * 32 * tap(iy, ix) + (32 - ay) * ax * 32 * tap(iy, ix + 1)
           + ay * (32 - ax) * 32 * tap(iy + 1, ix) + ay * ax * 32 * tap(iy + 1, ix + 1) + (1 << 14)) >> 15
    return np.clip(acc, 0, 255).astype(np.uint8)


def warp_flow(img: np.ndarray, flow: np.ndarray, binarize: bool = True) -> np.ndarray:
    res = remap_linear_u8(img, remap_map(flow))
    return np.equal(res, 1).astype(np.uint8) if binarize else res


def mask_iou(dt: List[np.ndarray], gt: List[np.ndarray]) -> np.ndarray:
    """pycocotools.mask.iou(dt_rles, gt_rles, [0]*len(gt)) on the masks themselves -> float64 [len(dt), len(gt)]."""
    out = np.zeros((len(dt), len(gt)), np.float64)
    for d, a in enumerate(dt):
        for g, b in enumerate(gt):
            i = int(np.count_nonzero((a != 0) & (b != 0)))
            u = int(np.count_nonzero((a != 0) | (b != 0))) if i else 1
            out[d, g] = float(i) / float(u)
    return out


def warp_proposals(proposals: List[Dict], flow: np.ndarray, rle_mod) -> List[Dict]:
    """merge_functions.py:219-241 with the flow already loaded; ``rle_mod`` supplies encode / to_bbox."""
    out = []
    for p in proposals:
        m = warp_flow(p["mask"], flow)
        seg = rle_mod.encode(m)
        out.append({"segmentation": seg, "bbox": rle_mod.to_bbox(seg), "score": 0.5 * (p["final_score"] + 1),
                    "final_score": p["final_score"], "object_score": p["object_score"], "mask": m, "id": p["id"]})
    return out
