"""Device-side counterparts of the mask work MergeTrack does per frame on the hot path's outputs:

  ``warp_flow`` / ``warp_proposals``   MergeTrack/merge_functions.py:209-241   (cv2.remap + == 1, RLE, bbox, scores)
  ``mask_iou``                         the pycocotools ``iou`` of merge_functions.py:38-45 on masks in HBM
  ``encode_masks``                     pycocotools ``encode(np.asfortranarray(mask))`` with ``counts`` as str

Same names, argument meaning and return shapes as the reference functions; masks may be numpy arrays (copied in) or
uint8 CUDA tensors (used in place).  Everything that touches pixels runs in libpremvos_hip.so (no CPU fallback);
only run-length differencing and the ASCII packing of ``counts`` happen on the host, on a few hundred integers.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Union

import numpy as np
import torch

from . import _lib, rle

ArrayLike = Union[np.ndarray, torch.Tensor]


def _dev_masks(masks, device="cuda") -> torch.Tensor:
    """-> uint8 [n,h,w] contiguous CUDA tensor."""
    if isinstance(masks, torch.Tensor):
        t = masks
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(m) for m in masks]) if isinstance(masks, (list, tuple))
                                                  else np.asarray(masks)))
    if t.dim() == 2:
        t = t.unsqueeze(0)
    assert t.dim() == 3, t.shape
    return t.to(device=device, dtype=torch.uint8).contiguous()


def get_flow(filename: str) -> np.ndarray:
    """merge_functions.py:197-207."""
    from .flow.driver import readFlowFile
    return readFlowFile(filename)


def warp_masks(masks: ArrayLike, flow: ArrayLike, binarize: bool = True) -> torch.Tensor:
    """All masks of a frame by one flow field, on the GPU: uint8 [n,h,w] -> uint8 [n,h,w] (stays in HBM)."""
    _lib.require_gpu()
    m = _dev_masks(masks)
    f = (flow if isinstance(flow, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(flow, dtype=np.float32)))
    f = f.to(device=m.device, dtype=torch.float32).contiguous()
    n, h, w = m.shape
    assert f.shape == (h, w, 2), (f.shape, m.shape)
    out = torch.empty_like(m)
    _lib.check(_lib.load().premvos_mask_warp_u8(m.data_ptr(), n, h, w, f.data_ptr(), out.data_ptr(), int(binarize),
                                                _lib.current_stream()), "mask_warp")
    return out


def warp_flow(img: np.ndarray, flow: np.ndarray, binarize: bool = True) -> np.ndarray:
    """merge_functions.py:209-217 for one uint8 image (numpy in, numpy out)."""
    return warp_masks(np.asarray(img)[None], flow, binarize)[0].cpu().numpy()


def mask_overlap(a: ArrayLike, b: ArrayLike):
    """-> (inter int64 [nb,na], area_a int64 [na], area_b int64 [nb]) as CUDA tensors."""
    _lib.require_gpu()
    ma, mb = _dev_masks(a), _dev_masks(b)
    assert ma.shape[1:] == mb.shape[1:], (ma.shape, mb.shape)
    na, nb = ma.shape[0], mb.shape[0]
    inter = torch.empty((nb, na), dtype=torch.int64, device=ma.device)
    aa = torch.empty((na,), dtype=torch.int64, device=ma.device)
    ab = torch.empty((nb,), dtype=torch.int64, device=ma.device)
    _lib.check(_lib.load().premvos_mask_overlap_u8(ma.data_ptr(), na, mb.data_ptr(), nb, ma.shape[1] * ma.shape[2],
                                                   inter.data_ptr(), aa.data_ptr(), ab.data_ptr(),
                                                   _lib.current_stream()), "mask_overlap")
    return inter, aa, ab


def mask_iou(dt: ArrayLike, gt: ArrayLike) -> np.ndarray:
    """``pycocotools.mask.iou(dt, gt, [0]*len(gt))`` on masks: float64 [len(dt), len(gt)]; 0 where disjoint."""
    inter, aa, ab = mask_overlap(dt, gt)
    i = inter.cpu().numpy().astype(np.float64).T                       # [na, nb]
    u = aa.cpu().numpy()[:, None].astype(np.float64) + ab.cpu().numpy()[None, :].astype(np.float64) - i
    return np.where(i > 0, i / np.where(i > 0, u, 1.0), 0.0)


def encode_masks(masks: ArrayLike) -> List[Dict[str, object]]:
    """COCO RLE of every mask ({"size": [h, w], "counts": str}); run boundaries are found on the GPU."""
    _lib.require_gpu()
    m = _dev_masks(masks)
    n, h, w = m.shape
    lib = _lib.load()
    cap = 4096
    ws = torch.empty((int(lib.premvos_rle_workspace_bytes(n, h, w)) + 3) // 4, dtype=torch.int32, device=m.device)
    nruns = torch.empty((n,), dtype=torch.int32, device=m.device)
    while True:
        pos = torch.empty((n, cap), dtype=torch.int32, device=m.device)
        _lib.check(lib.premvos_rle_boundaries_u8(m.data_ptr(), n, h, w, pos.data_ptr(), cap, nruns.data_ptr(),
                                                 ws.data_ptr(), _lib.current_stream()), "rle_boundaries")
        cnt = nruns.cpu().numpy()
        if int(cnt.max()) <= cap:
            break
        cap = int(cnt.max())                      # a very ragged mask: one retry with the exact capacity
    posh = pos.cpu().numpy()
    out = []
    for i in range(n):
        edges = np.concatenate([[0], posh[i, :cnt[i]].astype(np.int64), [h * w]])
        out.append({"size": [int(h), int(w)], "counts": rle.counts_to_string(np.diff(edges))})
    return out


def warp_proposals(proposals: Sequence[Dict], optflow: Union[str, ArrayLike]) -> List[Dict]:
    """merge_functions.py:219-241: every proposal's 'mask' warped to the next frame, with its RLE, bbox and scores.
    ``optflow`` is a .flo filename (as in the reference) or a flow array / CUDA tensor [h,w,2]."""
    flow = get_flow(optflow) if isinstance(optflow, str) else optflow
    if not proposals:
        return []
    warped = warp_masks([p["mask"] for p in proposals], flow)
    segs = encode_masks(warped)
    wm = warped.cpu().numpy()
    out = []
    for i, p in enumerate(proposals):
        out.append({"segmentation": segs[i], "bbox": rle.to_bbox(segs[i]), "score": 0.5 * (p["final_score"] + 1),
                    "final_score": p["final_score"], "object_score": p["object_score"], "mask": wm[i], "id": p["id"]})
    return out


# =====================================================================================================================
# The merge stage itself (MergeTrack/merge.py:71-136 + the host-side functions of merge_functions.py): sequential in t,
# calling the hot path's in-process services (refinement + ReID) on the warped proposals of every frame.  Pure
# bookkeeping / small-matrix numpy on the host; every per-pixel operation goes through the kernels above.
# =====================================================================================================================
import glob as _glob
import json as _json
import os as _os
from copy import deepcopy as _copy

MAX_REID_DISTANCE = 25                                            # merge_functions.py:12
SCORE_THRESH = 1e-10                                              # merge.py:123
WEIGHTS = np.array([0.25920137, 0.22541801, 0.0775609, 0.12509281, 0.3127269])      # merge.py:125


def pascal_colormap() -> np.ndarray:
    """The 256-entry PASCAL VOC palette (what merge_functions.py:250-506 tabulates as /255 floats) -> uint8 [256,3]."""
    cm = np.zeros((256, 3), np.uint8)
    for i in range(256):
        c, r, g, b = i, 0, 0, 0
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        cm[i] = (r, g, b)
    return cm


def save_with_pascal_colormap(filename: str, arr: np.ndarray) -> None:
    """merge_functions.py:508-514: palette PNG whose indices are the object ids."""
    from PIL import Image
    im = Image.fromarray(np.squeeze(arr).astype(np.uint8), mode="P")
    im.putpalette(pascal_colormap().reshape(-1).tolist())
    im.save(filename)


def save_pngs(proposals: Sequence[Dict], output_fn: str, empty: bool = False) -> None:
    """merge_functions.py:516-525."""
    png = np.zeros_like(proposals[0]["mask"])
    if not empty:
        for prop in proposals:
            png[prop["mask"].astype(bool)] = prop["id"]
    _os.makedirs(_os.path.dirname(output_fn) or ".", exist_ok=True)
    save_with_pascal_colormap(output_fn, png)


def read_ann(ann_fn: str) -> List[Dict]:
    """merge_functions.py:14-25: first-frame annotation PNG -> one template per object id."""
    from PIL import Image
    ann = np.array(Image.open(ann_fn))
    out = []
    for id_ in [i for i in np.unique(ann) if i != 0]:
        seg = rle.encode((ann == id_).astype(np.uint8))
        out.append({"id": id_, "bbox": np.array(rle.to_bbox(seg), np.float64), "segmentation": seg, "conf_score": "1.0",
                    "score": 1.0})
    return out


def read_props(prop_fn: str) -> List[Dict]:
    """merge_functions.py:27-36 (an unreadable file is an empty list; proposals without 'ReID' get an infinite one)."""
    try:
        with open(prop_fn, "r") as f:
            proposals = _json.load(f)
        for prop in proposals:
            if "ReID" not in prop:
                prop["ReID"] = np.inf * np.ones((128))
    except Exception:                                         # noqa: BLE001 -- the reference's bare except
        proposals = []
    return proposals


def calculate_scores(proposals: Sequence[Dict], templates: Sequence[Dict]) -> np.ndarray:
    """merge_functions.py:38-76 -> float64 [5, n_templates, n_proposals]:
    (objectness, ReID, 1 - best other ReID, warp IoU x template weight, 1 - best other warp)."""
    pm = np.stack([rle.decode(p["segmentation"]) for p in proposals])
    tm = np.stack([rle.decode(t["segmentation"]) for t in templates])
    warp_scores = mask_iou(pm, tm).T                                              # [templates, proposals]
    w = np.array([t["score"] for t in templates], np.float64)[:, np.newaxis]
    warp_scores = warp_scores * (np.maximum(w - 0.5, 0) / (1 - 0.5))
    dist = np.array([[np.linalg.norm(np.array(p["ReID"]) - np.array(t["ReID"])) for p in proposals] for t in templates])
    with np.errstate(invalid="ignore"):
        reid = 1 - dist / MAX_REID_DISTANCE
        reid[np.isinf(reid)] = 0
        reid[np.less(reid, 0)] = 0                  # (inf - inf = nan stays nan, as in the reference; zeroed on selection)
    other_warp, other_reid = np.ones_like(warp_scores), np.ones_like(reid)
    if len(templates) > 1:
        ids = np.arange(len(templates))
        for i in ids:
            other_warp[i, :] = 1 - np.max(np.atleast_2d(warp_scores[ids != i, :]), axis=0)
            other_reid[i, :] = 1 - np.max(np.atleast_2d(reid[ids != i, :]), axis=0)
    mask_scores = np.repeat(np.array([float(p["score"]) for p in proposals])[np.newaxis, :], warp_scores.shape[0], axis=0)
    mask_scores = np.maximum(mask_scores - 0.5, 0) / (1 - 0.5)
    return np.array([mask_scores, reid, other_reid, warp_scores, other_warp])


def calculate_selected_props(proposals: List[Dict], weighted_scores: np.ndarray, templates: Sequence[Dict],
                             score_thresh: float, object_scores: np.ndarray) -> List[Dict]:
    """merge_functions.py:96-121: arg-max proposal per template, an empty proposal when nothing beats the threshold."""
    h, w = proposals[0]["segmentation"]["size"]
    empty_seg = rle.encode(np.zeros((h, w), np.uint8))
    proposals.append({"segmentation": empty_seg, "bbox": np.array(rle.to_bbox(empty_seg), np.float64)})
    ws = np.append(weighted_scores, score_thresh * np.ones((weighted_scores.shape[0], 1)), axis=1)
    ws[np.logical_not(np.isfinite(ws))] = 0
    best, best_idx, best_obj = ws.max(axis=1), ws.argmax(axis=1), object_scores.max(axis=1)
    selected = [proposals[i].copy() for i in best_idx]
    for prop, score, template, obj in zip(selected, best, templates, best_obj):
        prop["final_score"], prop["object_score"], prop["id"] = score, obj, template["id"]
    return selected


def remove_mask_overlap(proposals: Sequence[Dict]) -> List[Dict]:
    """merge_functions.py:123-149: paint in ascending score so the best proposal wins every contested pixel."""
    scores = [p["final_score"] if p["final_score"] else 0 for p in proposals]
    masks = [rle.decode(p["segmentation"]) for p in proposals]
    order = np.argsort(scores)[::-1]
    labels = np.arange(1, len(order) + 1)
    png = np.zeros_like(masks[0])
    for i in order[::-1]:
        png[masks[i].astype(bool)] = labels[i]
    out = []
    for i, p in enumerate(proposals):
        m = (png == labels[i]).astype(np.uint8)
        seg = rle.encode(m)
        out.append({"segmentation": seg, "bbox": np.array(rle.to_bbox(seg), np.float64), "final_score": scores[i],
                    "object_score": p["object_score"] if p["object_score"] else 0, "mask": m, "id": p["id"]})
    return out


def update_templates(templates: Sequence[Dict], next_props: Sequence[Dict]) -> List[Dict]:
    """merge_functions.py:243-248."""
    new = _copy(list(next_props))
    for prop, template in zip(new, templates):
        prop["ReID"], prop["id"] = template["ReID"], template["id"]
    return new


def do_video(video_fn: str, refinement_net, ReID_net, input_images: str, first_frame_anns: str, input_proposals: str,
             input_optical_flow: str, output_images: str) -> List:
    """merge.py:71-121 for one sequence directory (``video_fn`` ends with '/').  Returns [(image_fn, selected_props)]."""
    from .refinement.driver import do_refinement
    from .reid.driver import add_ReID
    normalised_weights = WEIGHTS / np.sum(WEIGHTS)
    final_solution, templates, next_props = [], [], []
    image_fn_list = sorted(_glob.glob(video_fn + "*"))
    for image_id, image_fn in enumerate(image_fn_list):
        ann_fn = image_fn.replace(input_images, first_frame_anns).replace(".jpg", ".png")
        if _glob.glob(ann_fn) and "00000.jpg" in image_fn:
            new_templates = add_ReID(read_ann(ann_fn), image_fn, ReID_net)
            templates = templates + _copy(new_templates)
            next_props = next_props + _copy(new_templates)
        output_image_fn = image_fn.replace(input_images, output_images).replace(".jpg", ".png")
        if templates:
            prop_fn = image_fn.replace(input_images, input_proposals).replace(".jpg", ".json")
            proposals = next_props + read_props(prop_fn)
            all_scores = calculate_scores(proposals, templates)
            weighted_scores = np.dot(normalised_weights, all_scores.transpose((1, 0, 2)))
            object_scores = np.dot(np.array([1, 1]), all_scores[:2, :, :].transpose((1, 0, 2)))
            selected = calculate_selected_props(proposals, weighted_scores, templates, SCORE_THRESH, object_scores)
            selected = remove_mask_overlap(selected)
            optflow_fn = image_fn.replace(input_images, input_optical_flow).replace(".jpg", ".flo")
            if _glob.glob(optflow_fn):
                next_image_fn = image_fn_list[image_id + 1]
                next_props = warp_proposals(selected, optflow_fn)
                next_props = do_refinement(next_props, next_image_fn, refinement_net)
                next_props = add_ReID(next_props, next_image_fn, ReID_net)
                templates = update_templates(templates, next_props)
            final_solution.append((image_fn, selected))
            save_pngs(selected, output_image_fn)
        else:
            from PIL import Image
            h, w = np.asarray(Image.open(image_fn)).shape[:2]
            save_pngs([{"mask": np.zeros((h, w), np.uint8)}], output_image_fn, empty=True)
    return final_solution


def merge_all(refinement_net, ReID_net, root: str = ".", input_images: str = "data/DAVIS/JPEGImages/480p/",
              first_frame_anns: str = "data/DAVIS/Annotations/480p/",
              input_proposals: str = "output/intermediate/ReID_proposals/",
              input_optical_flow: str = "output/intermediate/flow/", output_images: str = "output/final/") -> int:
    """merge.py:129-136: every sequence that has proposals (paths relative to the PReMVOS root instead of code/)."""
    j = lambda p: _os.path.join(root, p)                      # noqa: E731
    videos = [v.replace(j(input_proposals), j(input_images)) for v in sorted(_glob.glob(j(input_proposals) + "*/"))]
    for v in videos:
        do_video(v, refinement_net, ReID_net, j(input_images), j(first_frame_anns), j(input_proposals),
                 j(input_optical_flow), j(output_images))
    return len(videos)
