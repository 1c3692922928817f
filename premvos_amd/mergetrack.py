"""Device-side counterparts of the mask work MergeTrack does per frame on the hot path's outputs.
MergeTrack itself (merge.py: do_video, scoring, selection, PNG output) is OUT OF SCOPE -- it keeps running unchanged and
may call these in place of cv2.remap / pycocotools:

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


def _dev_masks(masks, device=None) -> torch.Tensor:
    """-> uint8 [n,h,w] contiguous CUDA tensor: masks that are in HBM stay on THEIR device; host masks go to ``device`` (default: the
    calling thread's current GPU)."""
    if isinstance(masks, torch.Tensor):
        t = masks
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(m) for m in masks]) if isinstance(masks, (list, tuple))
                                                  else np.asarray(masks)))
    if t.dim() == 2:
        t = t.unsqueeze(0)
    assert t.dim() == 3, t.shape
    return t.to(device=t.device if t.is_cuda else _lib.resolve_device(device), dtype=torch.uint8).contiguous()


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


def encode_masks_begin(masks: ArrayLike):
    """First half of ``encode_masks``: the run boundaries of every mask are found on the GPU and copied to the host (a few
    hundred integers per mask).  Returns a handle for ``encode_masks_finish``, which needs neither the GPU nor the masks any
    more -- the drivers run it on the file-writer thread (FewShotSegmentationForwarder.py:140-141 does the whole encode inline)."""
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
        if n == 0 or int(cnt.max()) <= cap:
            break
        cap = int(cnt.max())                      # a very ragged mask: one retry with the exact capacity
    width = int(cnt.max()) if n else 0            # only the columns that hold boundaries cross PCIe
    return pos[:, :max(width, 1)].cpu().numpy(), cnt, int(h), int(w)


def encode_masks_finish(handle) -> List[Dict[str, object]]:
    posh, cnt, h, w = handle
    out = []
    for i in range(len(cnt)):
        edges = np.concatenate([[0], posh[i, :cnt[i]].astype(np.int64), [h * w]])
        out.append({"size": [h, w], "counts": rle.counts_to_string(np.diff(edges))})
    return out


def encode_masks(masks: ArrayLike) -> List[Dict[str, object]]:
    """COCO RLE of every mask ({"size": [h, w], "counts": str}); run boundaries are found on the GPU."""
    return encode_masks_finish(encode_masks_begin(masks))


def warp_proposals(proposals: Sequence[Dict], optflow: Union[str, ArrayLike], device_masks: bool = False) -> List[Dict]:
    """merge_functions.py:219-241: every proposal's 'mask' warped to the next frame, with its RLE, bbox and scores.
    ``optflow`` is a .flo filename (as in the reference) or a flow array / CUDA tensor [h,w,2].
    ``device_masks=True`` keeps the masks RESIDENT: the 'mask' entries of the result are uint8 CUDA tensors (views of one
    [n,h,w] tensor) instead of numpy arrays, and CUDA-tensor 'mask' entries are accepted on input, so a merge loop that feeds
    the result back into ``warp_proposals`` / ``mask_iou`` / ``encode_masks`` frame after frame never moves a mask over
    PCIe -- only the few hundred run boundaries of each RLE come to the host."""
    flow = get_flow(optflow) if isinstance(optflow, str) else optflow
    if not proposals:
        return []
    masks = [p["mask"] for p in proposals]
    if all(isinstance(m, torch.Tensor) and m.is_cuda for m in masks):
        masks = torch.stack([m.to(torch.uint8) for m in masks])
    warped = warp_masks(masks, flow)
    segs = encode_masks(warped)
    wm = warped if device_masks else warped.cpu().numpy()
    out = []
    for i, p in enumerate(proposals):
        out.append({"segmentation": segs[i], "bbox": rle.to_bbox(segs[i]), "score": 0.5 * (p["final_score"] + 1),
                    "final_score": p["final_score"], "object_score": p["object_score"], "mask": wm[i], "id": p["id"]})
    return out
