"""Generates tests/golden/merge_ref.npz / merge_host_refs.json by IMPORTING AND EXECUTING MergeTrack/merge_functions.py (unmodified)
in the build container: get_flow, warp_flow and warp_proposals (the GPU mask warp).  cv2 and pycocotools are absent from the image:

  * cv2.remap is a RECORDING stand-in: the fixture keeps the absolute sampling map the reference hands to it (that is what
    merge_functions.py:209-217 computes: -flow + pixel grid), the interpolation flag, and -- for the integer-valued flows used
    here, where bilinear interpolation is exact -- the gathered result; OpenCV's 1/32-pixel fixed-point interpolation itself
    stays restated (oracle/merge_oracle.py header);
  * pycocotools.mask.{encode, toBbox} are stand-ins on dense masks (encode keeps the mask, toBbox = tight x, y, w, h of the
    non-zero pixels): the fixture stores masks / boxes, never RLE strings.

Pins: the sign / grid convention of the warp, `== 1` binarisation, every key warp_proposals writes (score = 0.5 * (final_score
+ 1)).  Data only.
Usage: python tools/make_golden_merge.py [/root/reference]"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True

import tfshim  # noqa: E402  (only for its import stubs: tensorpack.utils.palette, scipy.misc)

tfshim.install()
CALLS = []


def remap(img, map1, map2, interpolation, *a, **k):
    assert map2 is None
    CALLS.append({"map": np.array(map1, np.float32), "interpolation": int(interpolation)})
    h, w = img.shape[:2]
    x, y = map1[..., 0], map1[..., 1]
    assert np.array_equal(x, np.rint(x)) and np.array_equal(y, np.rint(y)), "integer flows only: bilinear == gather"
    xi, yi = x.astype(np.int64), y.astype(np.int64)
    ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
    out = np.zeros(map1.shape[:2], img.dtype)
    out[ok] = img[yi[ok], xi[ok]]                                   # BORDER_CONSTANT 0 (the default of cv2.remap)
    return out


cv2 = types.ModuleType("cv2")
cv2.remap, cv2.INTER_LINEAR = remap, 1
sys.modules["cv2"] = cv2


def _bbox(m):
    ys, xs = np.nonzero(m)
    if len(ys) == 0:
        return np.zeros(4)
    return np.array([xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1], np.float64)


class _Seg(dict):
    """stands for an RLE dict: carries the dense mask; 'counts' is bytes-like as pycocotools returns it"""


def encode(m):
    s = _Seg(size=list(m.shape), counts=b"dense")
    s.mask = np.array(m, np.uint8)
    return s


def iou(dt, gt, iscrowd):
    out = np.zeros((len(dt), len(gt)))
    for i, a in enumerate(dt):
        for j, b in enumerate(gt):
            inter = np.count_nonzero(a.mask & b.mask)
            union = np.count_nonzero(a.mask | b.mask)
            out[i, j] = inter / union if union else 0.0
    return out


pm = types.ModuleType("pycocotools.mask")
pm.encode, pm.iou, pm.toBbox = encode, iou, lambda s: _bbox(s.mask)
pm.area, pm.decode, pm.merge = (lambda s: int(s.mask.sum())), (lambda s: s.mask), None
pc = types.ModuleType("pycocotools")
pc.mask = pm
sys.modules.update({"pycocotools": pc, "pycocotools.mask": pm})
sys.path.insert(0, os.path.join(REF, "code", "MergeTrack"))
import merge_functions as MF  # noqa: E402


def blobs(rng, h, w, n):
    m = np.zeros((h, w), np.uint8)
    for _ in range(n):
        cy, cx, ry, rx = rng.integers(5, h - 5), rng.integers(5, w - 5), rng.integers(3, h // 3), rng.integers(3, w // 3)
        yy, xx = np.mgrid[:h, :w]
        m[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] = 1
    return m


def main():
    rng = np.random.default_rng(31)
    h, w = 40, 56
    # an integer-valued flow field with regions moving differently (and out of the frame)
    flow = np.zeros((h, w, 2), np.float32)
    flow[:, : w // 2] = (3, -2)
    flow[:, w // 2:] = (-4, 5)
    flow[: h // 4] = (0, 9)
    masks = [blobs(rng, h, w, 2) for _ in range(3)]
    props = [{"mask": m, "id": int(i + 1), "final_score": float(fs), "object_score": float(os_)}
             for i, (m, fs, os_) in enumerate(zip(masks, (0.4, -0.2, 0.95), (0.9, 0.3, 0.6)))]
    with tempfile.TemporaryDirectory() as td:
        fn = os.path.join(td, "f.flo")
        with open(fn, "wb") as f:
            np.array([202021.25], np.float32).tofile(f)
            np.array([w, h], np.int32).tofile(f)
            flow.tofile(f)
        CALLS.clear()
        warped = MF.warp_proposals(props, fn)
    n_calls = len(CALLS)                                            # one cv2.remap per mask
    arrays = {"flow": flow, "masks": np.stack(masks), "warped_masks": np.stack([p["mask"] for p in warped]),
              "remap_map": CALLS[0]["map"], "warped_bbox": np.stack([np.asarray(p["bbox"], np.float64) for p in warped])}
    # warp_flow(binarize=False) on a grey image: the un-binarised gather
    img = rng.integers(0, 3, (h, w)).astype(np.uint8)
    arrays["grey"], arrays["grey_warped"] = img, MF.warp_flow(img, flow.copy(), binarize=False)
    arrays["grey_warped_bin"] = MF.warp_flow(img, flow.copy())
    g = {"remap_calls": n_calls, "interpolation_flag_is_INTER_LINEAR": all(c["interpolation"] == 1 for c in CALLS),
         "warped": [{k: (v if not isinstance(v, (np.floating, np.integer)) else v.item()) for k, v in p.items()
                     if k in ("score", "final_score", "object_score", "id")} for p in warped],
         "warped_keys": sorted(warped[0].keys()), "segmentation_counts_is_str": isinstance(warped[0]["segmentation"]["counts"], str)}
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, "merge_ref.npz"), **arrays)
    with open(os.path.join(GOLD, "merge_host_refs.json"), "w") as f:
        json.dump(g, f, indent=1)
    for fn in ("merge_ref.npz", "merge_host_refs.json"):
        print(fn, os.path.getsize(os.path.join(GOLD, fn)), "bytes")
    print(g["warped"])


if __name__ == "__main__":
    main()
