"""DAVIS-2017 semi-supervised evaluation measures, restated from the benchmark's published definition (Pont-Tuset et al., "The 2017
DAVIS Challenge on Video Object Segmentation", arXiv:1704.00675, section 3; Perazzi et al., CVPR 2016, section 4) so that the
reference's acceptance numbers (README.md:35-38: Mean J 0.7363, Mean F 0.80044, Mean J&F 76.8366 on DAVIS-2017 val) can be
computed on a box that has the weights and the dataset but not the davis2017-evaluation package:

  J  region similarity: |M and G| / |M or G| per object and frame (1 when both are empty)
  F  contour accuracy: F-measure of the boundary pixels of M and G matched within 0.008 x the image diagonal
     (bipartite matching approximated by morphological dilation, as in the official code)
  per object: mean over the frames of the sequence EXCLUDING the first and the last one; the benchmark numbers are the
  means over all objects of all sequences; J&F = (mean J + mean F) / 2.

Results and annotations are palette PNGs whose pixel values are object ids (0 = background): the format MergeTrack writes
(merge_functions.py:516-525) and DAVIS ships."""
from __future__ import annotations

import glob
import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
from scipy import ndimage


def db_eval_iou(annotation: np.ndarray, segmentation: np.ndarray) -> float:
    a, s = annotation.astype(bool), segmentation.astype(bool)
    union = np.logical_or(a, s).sum()
    if union == 0:
        return 1.0
    return float(np.logical_and(a, s).sum()) / float(union)


def seg2bmap(seg: np.ndarray) -> np.ndarray:
    """Boundary map of a binary mask: a pixel is a boundary pixel when its east, south or south-east neighbour differs (the last
    row / column compare with themselves)."""
    seg = seg.astype(bool)
    e = np.zeros_like(seg)
    s = np.zeros_like(seg)
    se = np.zeros_like(seg)
    e[:, :-1] = seg[:, 1:]
    s[:-1, :] = seg[1:, :]
    se[:-1, :-1] = seg[1:, 1:]
    b = (seg ^ e) | (seg ^ s) | (seg ^ se)
    b[-1, :] = seg[-1, :] ^ e[-1, :]
    b[:, -1] = seg[:, -1] ^ s[:, -1]
    b[-1, -1] = False
    return b


def _disk(r: int) -> np.ndarray:
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    return (x * x + y * y) <= r * r


def db_eval_boundary(segmentation: np.ndarray, annotation: np.ndarray, bound_th: float = 0.008) -> float:
    bound_pix = bound_th if bound_th >= 1 else int(np.ceil(bound_th * np.linalg.norm(segmentation.shape)))
    fg, gt = seg2bmap(segmentation), seg2bmap(annotation)
    disk = _disk(int(bound_pix))
    fg_dil = ndimage.binary_dilation(fg, structure=disk)
    gt_dil = ndimage.binary_dilation(gt, structure=disk)
    gt_match, fg_match = gt & fg_dil, fg & gt_dil
    n_fg, n_gt = int(fg.sum()), int(gt.sum())
    if n_fg == 0 and n_gt > 0:
        precision, recall = 1.0, 0.0
    elif n_fg > 0 and n_gt == 0:
        precision, recall = 0.0, 1.0
    elif n_fg == 0 and n_gt == 0:
        precision, recall = 1.0, 1.0
    else:
        precision, recall = float(fg_match.sum()) / n_fg, float(gt_match.sum()) / n_gt
    return 0.0 if precision + recall == 0 else 2.0 * precision * recall / (precision + recall)


def _read_ids(fn: str) -> np.ndarray:
    from PIL import Image
    im = Image.open(fn)
    a = np.asarray(im)
    if a.ndim == 3:                               # an RGB rendering instead of a palette image: map colours to ids by first use
        flat = a.reshape(-1, a.shape[-1])
        cols, inv = np.unique(flat, axis=0, return_inverse=True)
        order = np.argsort([0 if not c.any() else 1 for c in cols], kind="stable")
        remap = np.empty(len(cols), np.int64)
        remap[order] = np.arange(len(cols))
        a = remap[inv].reshape(a.shape[:2])
    return a.astype(np.int64)


def evaluate_sequence(result_dir: str, annotation_dir: str) -> Dict[int, Tuple[float, float]]:
    """{object id: (mean J, mean F)} of one sequence; frames without a result file count as empty masks."""
    ann_files = sorted(glob.glob(os.path.join(annotation_dir, "*.png")))
    if len(ann_files) < 3:
        raise ValueError(f"{annotation_dir}: need at least three annotated frames")
    first = _read_ids(ann_files[0])
    ids = [int(i) for i in np.unique(first) if i != 0]
    per: Dict[int, Tuple[List[float], List[float]]] = {i: ([], []) for i in ids}
    for fn in ann_files[1:-1]:                                    # first and last frame are excluded (semi-supervised protocol)
        gt = _read_ids(fn)
        rf = os.path.join(result_dir, os.path.basename(fn))
        res = _read_ids(rf) if os.path.exists(rf) else np.zeros_like(gt)
        if res.shape != gt.shape:
            raise ValueError(f"{rf}: shape {res.shape} differs from the annotation's {gt.shape}")
        for i in ids:
            per[i][0].append(db_eval_iou(gt == i, res == i))
            per[i][1].append(db_eval_boundary(res == i, gt == i))
    return {i: (float(np.mean(j)), float(np.mean(f))) for i, (j, f) in per.items()}


def evaluate(results_root: str, annotations_root: str, sequences: Optional[List[str]] = None) -> dict:
    """results_root/<seq>/<frame>.png against annotations_root/<seq>/<frame>.png (DAVIS: Annotations/480p)."""
    seqs = sequences or sorted(d for d in os.listdir(annotations_root) if os.path.isdir(os.path.join(results_root, d)))
    js, fs, table = [], [], {}
    for s in seqs:
        r = evaluate_sequence(os.path.join(results_root, s), os.path.join(annotations_root, s))
        table[s] = {str(i): {"J": round(j, 5), "F": round(f, 5)} for i, (j, f) in r.items()}
        js += [j for j, _ in r.values()]
        fs += [f for _, f in r.values()]
    mj, mf = (float(np.mean(js)) if js else math.nan), (float(np.mean(fs)) if fs else math.nan)
    return {"mean_J": round(mj, 5), "mean_F": round(mf, 5), "mean_JF_percent": round(50.0 * (mj + mf), 4), "objects": len(js),
            "sequences": len(seqs), "per_sequence": table}
