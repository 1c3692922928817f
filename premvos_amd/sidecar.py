"""Optional binary side-car for per-frame proposals: the same content as the proposal JSON of the
reference -- ``[{"bbox": [x, y, w, h], "score": s, "segmentation": {"size", "counts"}, "conf_score": str, "ReID": [...]}, ...]``
(train.py:399-427, FewShotSegmentationForwarder.py:139-155, Forwarding/ReIDForwarding.py:34-92) -- with the masks as packed
bits instead of COCO-RLE strings.  It exists for OUR consumers (the ReID stage driver, the GPU mask helpers of
``premvos_amd.mergetrack``): between two stages of this package the RLE string of every mask is written by one stage only to be
parsed back by the next.  JSON + RLE stays the default and the only format MergeTrack / the reference's ReID stage read;
``to_proposals`` / ``python -m premvos_amd.sidecar --to-json`` turn a side-car back into exactly that JSON.

File layout (little endian):  b"PMVB", u32 version = 1, u32 H, u32 W, u32 n, u32 flags (1 = masks, 2 = conf_score, 4 = ReID),
f64 bbox[n][4], f64 score[n], [f32 conf[n]], [u8 bits[n][ceil(H*W/8)]  (row-major pixels, bit k of a byte = pixel 8*byte + k)],
[f32 reid[n][128]], u8 has_reid[n] when flag 4.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import struct
import sys
from typing import Dict, List, Optional

import numpy as np

from . import rle

MAGIC, VERSION = b"PMVB", 1
F_MASKS, F_CONF, F_REID = 1, 2, 4
EXT = ".pmv"


def write(fn: str, height: int, width: int, bbox, score, conf=None, mask_bits=None, reid=None, has_reid=None) -> None:
    bbox = np.ascontiguousarray(bbox, np.float64).reshape(-1, 4)
    n = len(bbox)
    score = np.ascontiguousarray(score, np.float64).reshape(n)
    flags = (F_MASKS if mask_bits is not None else 0) | (F_CONF if conf is not None else 0) | (F_REID if reid is not None else 0)
    nbytes = (height * width + 7) // 8
    with open(fn, "wb") as f:
        f.write(MAGIC + struct.pack("<5I", VERSION, height, width, n, flags))
        f.write(bbox.tobytes())
        f.write(score.tobytes())
        if conf is not None:
            f.write(np.ascontiguousarray(conf, np.float32).reshape(n).tobytes())
        if mask_bits is not None:
            mb = np.ascontiguousarray(mask_bits, np.uint8).reshape(n, nbytes)
            f.write(mb.tobytes())
        if reid is not None:
            f.write(np.ascontiguousarray(reid, np.float32).reshape(n, 128).tobytes())
            f.write(np.ascontiguousarray(has_reid if has_reid is not None else np.ones(n), np.uint8).reshape(n).tobytes())


def read(fn: str) -> Dict[str, object]:
    with open(fn, "rb") as f:
        raw = f.read()
    if raw[:4] != MAGIC:
        raise ValueError(f"{fn}: not a proposal side-car (magic {raw[:4]!r})")
    version, h, w, n, flags = struct.unpack_from("<5I", raw, 4)
    if version != VERSION:
        raise ValueError(f"{fn}: side-car version {version}, this reader knows {VERSION}")
    off = 24
    nbytes = (h * w + 7) // 8

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype, count, off)
        off += a.nbytes
        return a
    d: Dict[str, object] = {"height": h, "width": w, "bbox": take(np.float64, 4 * n).reshape(n, 4), "score": take(np.float64, n),
                            "conf": None, "mask_bits": None, "reid": None, "has_reid": None}
    if flags & F_CONF:
        d["conf"] = take(np.float32, n)
    if flags & F_MASKS:
        d["mask_bits"] = take(np.uint8, n * nbytes).reshape(n, nbytes)
    if flags & F_REID:
        d["reid"] = take(np.float32, n * 128).reshape(n, 128)
        d["has_reid"] = take(np.uint8, n)
    if off != len(raw):
        raise ValueError(f"{fn}: {len(raw) - off} trailing bytes")
    return d


def pack_masks(masks: np.ndarray) -> np.ndarray:
    """uint8 [n,H,W] (non-zero = foreground) -> bits [n, ceil(H*W/8)], the layout of premvos_mask_pack_bits_u8."""
    m = np.ascontiguousarray(masks).reshape(len(masks), -1) != 0
    return np.packbits(m, axis=1, bitorder="little")


def unpack_masks(d: Dict[str, object]) -> np.ndarray:
    h, w = d["height"], d["width"]
    bits = d["mask_bits"]
    return np.unpackbits(bits, axis=1, count=h * w, bitorder="little").reshape(len(bits), h, w)


def tight_bbox(mask: np.ndarray) -> List[float]:
    """pycocotools toBbox of the mask's RLE: [x, y, w, h] of the non-zero pixels, zeros for an empty mask."""
    return rle.to_bbox(rle.encode(mask))


def to_proposals(d: Dict[str, object]) -> List[dict]:
    """The list the JSON file of the same frame holds (same Python floats, RLE strings and conf_score strings)."""
    masks = unpack_masks(d) if d["mask_bits"] is not None else None
    out = []
    for i in range(len(d["bbox"])):
        p = {"bbox": [float(v) for v in d["bbox"][i]], "score": float(d["score"][i])}
        if masks is not None:
            p["segmentation"] = rle.encode(masks[i])
        if d["conf"] is not None:
            p["conf_score"] = str(d["conf"][i])               # str(np.float32), as FewShotSegmentationForwarder.py:148 stores it
        if d["reid"] is not None and d["has_reid"][i]:
            p["ReID"] = np.array(d["reid"][i]).tolist()
        out.append(p)
    return out


def from_proposals(props: List[dict], height: int, width: int) -> Dict[str, object]:
    """JSON-style proposals -> side-car arrays (masks from the RLE)."""
    n = len(props)
    d: Dict[str, object] = {"height": height, "width": width,
                            "bbox": np.array([p["bbox"] for p in props], np.float64).reshape(n, 4),
                            "score": np.array([p["score"] for p in props], np.float64), "conf": None, "mask_bits": None,
                            "reid": None, "has_reid": None}
    if n and all("conf_score" in p for p in props):
        d["conf"] = np.array([np.float32(p["conf_score"]) for p in props], np.float32)
    if n and all("segmentation" in p for p in props):
        d["mask_bits"] = pack_masks(np.stack([rle.decode(p["segmentation"]) for p in props]))
    if any("ReID" in p for p in props):
        d["has_reid"] = np.array(["ReID" in p for p in props], np.uint8)
        d["reid"] = np.stack([np.asarray(p.get("ReID", np.zeros(128)), np.float32) for p in props])
    return d


def write_dict(fn: str, d: Dict[str, object]) -> None:
    write(fn, d["height"], d["width"], d["bbox"], d["score"], d["conf"], d["mask_bits"], d["reid"], d["has_reid"])


def convert_tree(src: str, dst: str) -> int:
    """Every <src>/<seq>/<frame>.pmv -> <dst>/<seq>/<frame>.json (what MergeTrack and the reference's stages read)."""
    n = 0
    for fn in sorted(glob.glob(os.path.join(src, "*", "*" + EXT))):
        rel = os.path.splitext(os.path.relpath(fn, src))[0] + ".json"
        os.makedirs(os.path.dirname(os.path.join(dst, rel)), exist_ok=True)
        with open(os.path.join(dst, rel), "w") as f:
            f.write(json.dumps(to_proposals(read(fn))))
        n += 1
    return n


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(description="proposal side-car -> the reference's JSON")
    ap.add_argument("--to-json", nargs=2, metavar=("SRC_DIR", "DST_DIR"), required=True)
    a = ap.parse_args(argv)
    print("files:", convert_tree(*a.to_json))
    return 0


if __name__ == "__main__":
    sys.exit(main())
