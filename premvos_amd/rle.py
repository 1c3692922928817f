"""COCO run-length masks: the on-disk 'segmentation' format the ReID and MergeTrack stages consume
(refinement_net/forwarding/FewShotSegmentationForwarder.py:139-143 writes pycocotools.mask.encode output with
'counts' decoded to str; ReID_net/datasets/Similarity/DAVIS_Forward_Similarity.py:38-39 reads it via toBbox).

pycocotools is not a dependency here; the format (maskApi.c) is: column-major run lengths alternating 0/1 and
starting with zeros; ``counts`` string = each run (delta-coded against the run two back, from the 4th on) in
5-bit groups, LSB first, bit 0x20 = continuation, +48 to make it ASCII.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np


def counts_from_mask(mask: np.ndarray) -> np.ndarray:
    flat = (np.asarray(mask) != 0).reshape(-1, order="F")
    if flat.size == 0:
        return np.zeros((0,), np.int64)
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    runs = np.diff(np.concatenate(([0], change, [flat.size])))
    if flat[0]:
        runs = np.concatenate(([0], runs))
    return runs.astype(np.int64)


def counts_to_string(counts) -> str:
    if len(counts) > 64:              # long run lists: the C twin in libpremvos_hip.so (same algorithm, no GPU involved)
        try:
            import ctypes as C
            from . import _lib
            arr = np.ascontiguousarray(counts, dtype=np.int64)
            buf = C.create_string_buffer(13 * len(arr) + 1)
            n = _lib.load().premvos_rle_counts_to_string_host(arr.ctypes.data, len(arr), buf, len(buf))
            if n >= 0:
                return buf.raw[:n].decode("ascii")
        except Exception:                  # noqa: BLE001 -- library not built yet: the pure-python path below is equivalent
            pass
    out = bytearray()
    cl = [int(c) for c in counts]
    for i, x in enumerate(cl):
        if i > 2:
            x -= cl[i - 2]
        while True:
            c = x & 0x1F
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(c + 48)
            if not more:
                break
    return out.decode("ascii")


def strings_from_pool(pool: np.ndarray, offsets: np.ndarray, hw: int) -> List[str]:
    """The ``counts`` strings of len(offsets) - 1 masks from their POOLED run boundaries (premvos_rle_boundaries_pooled_u8 after the
    copy to the host: mask i's ascending column-major change positions are pool[offsets[i]:offsets[i + 1]]; its run lengths are the
    successive differences of [0, positions..., hw]) -- one call into the C twin for all of them (interpreter lock released)."""
    import ctypes as C
    from . import _lib
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    n = len(offsets) - 1
    if n <= 0:
        return []
    first, last = int(offsets[0]), int(offsets[-1])
    sub = np.ascontiguousarray(pool[first:last], dtype=np.int32)
    rel = offsets - np.int32(first)
    cap = 13 * (last - first + n) + 16                  # <= 13 characters per run (a 64-bit value in 5-bit groups)
    buf = C.create_string_buffer(cap)
    so = np.empty(n + 1, np.int64)
    total = _lib.load().premvos_rle_strings_host(sub.ctypes.data, rel.ctypes.data, n, int(hw), buf, cap, so.ctypes.data)
    if total < 0:
        raise RuntimeError("premvos_rle_strings_host: output buffer too small")
    text = buf.raw[:total].decode("ascii")
    return [text[so[i]:so[i + 1]] for i in range(n)]


def string_to_counts(s: str) -> List[int]:
    counts: List[int] = []
    p, b = 0, s.encode("ascii")
    while p < len(b):
        x, k = 0, 0
        while True:
            c = b[p] - 48
            x |= (c & 0x1F) << (5 * k)
            p += 1
            k += 1
            if not (c & 0x20):
                if c & 0x10:
                    x |= -1 << (5 * k)
                break
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def encode(mask: np.ndarray) -> Dict[str, object]:
    """Equivalent of ``pycocotools.mask.encode(np.asfortranarray(mask))`` with counts as str."""
    h, w = mask.shape[:2]
    return {"size": [int(h), int(w)], "counts": counts_to_string(counts_from_mask(mask))}


def decode(rle: Dict[str, object]) -> np.ndarray:
    h, w = rle["size"]
    counts = string_to_counts(rle["counts"]) if isinstance(rle["counts"], str) else list(rle["counts"])
    ends = np.cumsum(counts)
    flat = np.zeros(h * w, np.uint8)
    for i in range(1, len(counts), 2):
        flat[ends[i - 1]:ends[i]] = 1
    return flat.reshape((h, w), order="F")


def area(rle: Dict[str, object]) -> int:
    c = string_to_counts(rle["counts"])
    return int(sum(c[1::2]))


def to_bbox(rle: Dict[str, object]) -> List[float]:
    """maskApi.c rleToBbox: [x, y, w, h] of the foreground (zeros if empty)."""
    h, w = rle["size"]
    c = string_to_counts(rle["counts"])
    if len(c) < 2 or sum(c[1::2]) == 0:
        return [0.0, 0.0, 0.0, 0.0]
    xs, ys, xe, ye = w, h, 0, 0
    pos = 0
    for i, n in enumerate(c):
        if i % 2 == 1 and n > 0:
            a, b = pos, pos + n - 1
            xa, xb = a // h, b // h
            xs, xe = min(xs, xa), max(xe, xb)
            if xa < xb:
                ys, ye = 0, h - 1
            else:
                ys, ye = min(ys, a % h), max(ye, b % h)
        pos += n
    return [float(xs), float(ys), float(xe - xs + 1), float(ye - ys + 1)]
