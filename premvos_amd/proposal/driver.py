"""Proposal stage driver: the counterpart of proposal_net/train.py --forward and eval.py.

Reference boundary kept:
  * ``pred = OfflinePredictor(...)``; ``pred(resized_img) -> (final_boxes, final_probs, final_labels,
    final_posterior, second_final_labels, second_final_posterior)``        (train.py:52-62, 653-657)
  * ``detect_one_image(img, model_func)``                                  (eval.py:61-110)
  * ``convert_results_to_json`` -> [{"bbox": [x,y,w,h] (1 decimal), "score": (2 decimals)}]   (train.py:388-428)
  * ``forward(pred_func, output_folder, forward_dataset)``: per-frame JSON, skip if it exists (train.py:431-522)
  * CLI flags used by simple_run.sh:33,41: --forward --agnostic --second_head --forward_dataset --load --davis_name
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
from collections import namedtuple
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import _lib, jpeg, ops
from .model import (RESNET_NUM_BLOCK, RESULTS_PER_IM, TEST_POST_NMS_TOPK, ProposalNet)

SHORT_EDGE_SIZE, MAX_SIZE = 800, 1333      # config.py:64-65

SecondDetectionResult = namedtuple(
    "SecondDetectionResult",
    ["box", "score", "class_id", "posterior", "mask", "second_class_id", "second_posterior", "feature_fastrcnn_pooled"])


def custom_resize_shape(h: int, w: int, size: int = SHORT_EDGE_SIZE, max_size: int = MAX_SIZE):
    """CustomResize._get_augment_params (common.py:47-62)."""
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


class OfflinePredictor:
    """Callable with the reference predictor's call shape: resized HWC BGR image -> tuple of numpy arrays."""

    def __init__(self, net: ProposalNet):
        self.net = net

    def __call__(self, resized_img: np.ndarray):
        img = np.ascontiguousarray(resized_img)
        if img.dtype != np.uint8:
            img = np.clip(np.rint(img), 0, 255).astype(np.uint8)     # the pipeline only ever feeds uint8 frames
        t = torch.from_numpy(img).unsqueeze(0).to(self.net.device)
        p = self.net.run_resized(t)
        out = self.net.outputs(p, 0)[:6]
        return out + (self.net.masks(p, 0),) if self.net.mode_mask else out   # + 'final_masks' (train.py:58-59)


class ProposalStage:
    """Raw uint8 BGR frames [B,H,W,3] on the device -> device-resident detections, resize fused on the GPU."""

    def __init__(self, weights: Dict[str, object], batch: int = 1, device=None,
                 num_blocks: Sequence[int] = RESNET_NUM_BLOCK, net: Optional[ProposalNet] = None,
                 use_graph: bool = True, rgb_input: bool = False, precision: Optional[str] = None):
        self.net = net if net is not None else ProposalNet(weights, num_blocks, device, use_graph=False,
                                                           precision=precision)
        self.batch, self.device, self.use_graph = batch, (self.net.device if net is not None and device is None else _lib.resolve_device(device)), use_graph
        self.rgb_input = rgb_input
        self._shape = None

    def _prepare(self, h: int, w: int):
        if self._shape == (h, w):
            return
        with ops.BUILD_LOCK:
            self._prepare_locked(h, w)

    def _prepare_locked(self, h: int, w: int):
        self.nh, self.nw = custom_resize_shape(h, w)
        self.scale = (self.nh * 1.0 / h + self.nw * 1.0 / w) / 2          # eval.py:78
        self.plan = self.net.plan(self.batch, self.nh, self.nw)
        self.frames = torch.empty((self.batch, h, w, 3), dtype=torch.uint8, device=self.device)
        lib, p, b = _lib.load(), self.plan, self.batch

        def pre():
            _lib.check(lib.premvos_proposal_preprocess_u8(self.frames.data_ptr(), b, h, w, p.img.ptr, self.nh, self.nw,
                                                          int(self.rgb_input), _lib.current_stream()),
                       "proposal_preprocess")
        self.steps = [("proposal_preprocess", pre)] + list(p.steps)
        self.graph = p.capture(self.steps) if self.use_graph else None
        self._shape = (h, w)

    def run(self, frames_bgr: torch.Tensor):
        """Returns the plan (final_boxes [B,20,4] in RESIZED-image coordinates, final_probs, final_count ...)."""
        assert frames_bgr.dtype == torch.uint8 and frames_bgr.shape[0] == self.batch
        self._prepare(frames_bgr.shape[1], frames_bgr.shape[2])
        self.frames.copy_(frames_bgr)
        if self.graph is not None:
            self.graph.replay()
        else:
            self.plan.run(self.steps)
        return self.plan

    def json_results(self, orig_hw) -> List[List[dict]]:
        """What ``forward`` writes for every image of the last batch: convert_results_to_json(detect_one_image(...)) with
        ONE device-to-host copy per output tensor for the whole batch (the per-image ``detections`` fetches the posteriors and
        second-head outputs too, seven small synchronous copies per image)."""
        assert not self.net.mode_mask
        p = self.plan
        return results_json(p.final_boxes.cpu().numpy(), p.final_probs.cpu().numpy(), p.final_count.cpu().numpy(), self.scale, orig_hw)

    def detections(self, i: int, orig_hw) -> List[SecondDetectionResult]:
        boxes, probs, labels, post, sl, sp, _ = self.net.outputs(self.plan, i)
        masks = self.net.masks(self.plan, i) if self.net.mode_mask else None
        return _to_results(boxes, probs, labels, post, sl, sp, self.scale, orig_hw, masks)


def results_json(boxes: np.ndarray, probs: np.ndarray, counts: np.ndarray, scale: float, orig_hw) -> List[List[dict]]:
    """The proposal JSON of a batch from the net's output arrays (``final_boxes`` [B,20,4] in RESIZED-image coordinates,
    ``final_probs`` [B,20], ``final_count`` [B]): un-scale, clip (eval.py:93-94), xywh / rounding (train.py:388-428).  A pure
    function of the arrays, so the merge rank of a gathered job writes the bytes the producing rank would have written."""
    out = []
    for i in range(len(counts)):
        n = int(counts[i])
        none = [None] * n
        res = _to_results(np.array(boxes[i, :n], dtype=np.float32), np.array(probs[i, :n], dtype=np.float32), np.ones((n,), np.int64), none, none,
                          none, scale, orig_hw)
        out.append(convert_results_to_json(res))
    return out


def _cv_resize_f32(img: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """cv2.resize(float32, INTER_LINEAR) on the host (eval.py:54 runs it on the CPU too): half-pixel centres,
    edge clamp, horizontal then vertical lerp (OpenCV imgproc resize.cpp)."""
    def co(dst, src):
        f = ((np.arange(dst, dtype=np.float64) + 0.5) * (1.0 / (float(dst) / float(src))) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo, hi = s < 0, s >= src - 1
        f[lo], s[lo] = 0.0, 0
        f[hi], s[hi] = 0.0, src - 1
        return s, np.minimum(s + 1, src - 1), f
    h, w = img.shape
    x0, x1, fx = co(dst_w, w)
    y0, y1, fy = co(dst_h, h)
    rows = img[:, x0] * (np.float32(1) - fx)[None, :] + img[:, x1] * fx[None, :]
    return (rows[y0] * (np.float32(1) - fy)[:, None] + rows[y1] * fy[:, None]).astype(np.float32)


def fill_full_mask(box, mask, shape) -> np.ndarray:
    """The mask head's M x M probabilities of one detection pasted into a frame-sized uint8 mask (eval.py:35-58; MODE_MASK only).
    Pixel columns [floor(x0 + 0.5), floor(x1 - 0.5)] (at least one) and the rows likewise receive the probabilities resized to
    that extent (cv2.resize INTER_LINEAR on float32) and thresholded at 0.5; the part of the box that lies beyond the frame's
    right / bottom edge is dropped."""
    corner = np.asarray(box)                     # (the +- 0.5 in the box's own dtype, then int() = truncation, as eval.py:47-49)
    first = np.trunc(corner[:2] + 0.5).astype(np.int64)
    last = np.maximum(np.trunc(corner[2:] - 0.5).astype(np.int64), first)
    (x_lo, y_lo), (cols, rows) = first, last - first + 1
    pasted = _cv_resize_f32(np.ascontiguousarray(mask, np.float32), int(cols), int(rows)) > 0.5
    frame = np.zeros(shape, dtype=np.uint8)
    keep_r, keep_c = max(0, min(int(rows), shape[0] - int(y_lo))), max(0, min(int(cols), shape[1] - int(x_lo)))
    frame[y_lo:y_lo + keep_r, x_lo:x_lo + keep_c] = pasted[:keep_r, :keep_c]
    return frame


def clip_boxes(boxes: np.ndarray, shape) -> np.ndarray:
    """Boxes (x0, y0, x1, y1 in the last axis) limited to an image of ``shape`` = (height, width): the top-left corner to
    >= 0, x1 to <= width, y1 to <= height -- one-sided, as the reference does it (common.py:107-119 leaves x0 / y0 above the
    frame and negative x1 / y1 alone).  Works in place on a C-contiguous input (eval.py:94 relies on that) and returns it."""
    flat = boxes.reshape(-1, 4)                 # a view for contiguous input: the clamps below write through to ``boxes``
    corner, far = flat[:, :2], flat[:, 2:]
    # (written with comparisons, not np.maximum / np.minimum: which zero maximum(-0.0, 0) returns depends on the SIMD path numpy picks
    #  for the host, and the C twin of this arithmetic -- csrc/host_files.hip -- must write the same "0.0"; NaN stays NaN)
    corner[corner <= 0] = 0
    limit = np.broadcast_to(np.asarray((shape[1], shape[0]), dtype=flat.dtype), far.shape)
    over = far > limit
    far[over] = limit[over]
    return flat.reshape(boxes.shape)


def _to_results(boxes, probs, labels, posteriors, second_labels, second_posteriors, scale, orig_shape, masks=None):
    boxes = boxes / scale                      # eval.py:93
    boxes = clip_boxes(boxes, orig_shape)      # eval.py:94
    full = [None] * len(boxes) if masks is None else [fill_full_mask(b, m, orig_shape) for b, m in zip(boxes, masks)]
    return [SecondDetectionResult(*a) for a in zip(boxes, probs, labels, posteriors, full,
                                                   second_labels, second_posteriors, [None] * len(boxes))]


def detect_one_image(img: np.ndarray, model_func) -> List[SecondDetectionResult]:
    """eval.py:61-110.  ``img`` HWC BGR uint8.  The resize runs on the GPU (bit-exact restatement of the
    cv2 fixed-point INTER_LINEAR) when ``model_func`` is our OfflinePredictor; a foreign callable gets a
    host-resized image exactly like the reference."""
    orig_shape = img.shape[:2]
    if isinstance(model_func, OfflinePredictor):
        stage = _stage_for(model_func.net)
        t = torch.from_numpy(np.ascontiguousarray(img[:, :, :3])).unsqueeze(0).to(stage.device)
        stage.run(t)
        return stage.detections(0, orig_shape)
    raise TypeError("model_func must be a premvos_amd OfflinePredictor")


_STAGES: Dict[tuple, ProposalStage] = {}


def _stage_for(net: ProposalNet, batch: int = 1) -> ProposalStage:
    if (id(net), batch) not in _STAGES:
        _STAGES[(id(net), batch)] = ProposalStage({}, batch=batch, device=net.device, net=net)
    return _STAGES[(id(net), batch)]


def convert_results_to_json(results, img_idx=None) -> List[dict]:
    """train.py:388-428 (MODE_MASK False in --forward: only bbox + score)."""
    img_res = []
    for r in results:
        box = np.array(r.box)
        box[2] -= box[0]
        box[3] -= box[1]
        res = {"bbox": list(map(lambda x: float(round(x, 1)), box)), "score": float(round(r.score, 2))}
        if r.mask is not None:                 # train.py:419-425 (MODE_MASK only)
            from .. import rle
            res["segmentation"] = rle.encode(r.mask)
        img_res.append(res)
    return img_res


def forward(pred_func, output_folder: str, forward_dataset: str, davis_name: Optional[str] = None,
            generic_images_folder: Optional[str] = None, generic_images_pattern: Optional[str] = None) -> int:
    """train.py:431-522: one JSON per frame under <output>/<seq>/<frame>.json; existing files are skipped."""
    from PIL import Image
    if forward_dataset.lower() == "davis":      # train.py:441-456: sequence paths are relative to the list file
        pre = "/".join(davis_name.split("/")[:-1])
        with open(davis_name) as f:
            seqs = [ln.rstrip() for ln in f if ln.rstrip()]
        from .. import parallel
        seqs = parallel.my_videos(seqs)         # under torch.distributed.run: this rank's slice of the video list
        imgs = []
        for s in seqs:
            imgs += sorted(glob.glob(pre + "/" + s + "/*"))
    else:
        imgs = sorted(glob.glob(os.path.join(generic_images_folder, generic_images_pattern)))
    # Frames still to do, grouped into runs of equal size and pushed through the net DRIVER_BATCH at a time (the
    # reference forwards one frame per session.run; every frame is independent, so batching changes nothing but speed)
    todo = []
    for fn in imgs:
        seq = fn.split("/")[-2]
        out_dir = os.path.join(output_folder, seq)
        os.makedirs(out_dir, exist_ok=True)
        out_fn = os.path.join(out_dir, os.path.splitext(os.path.basename(fn))[0] + ".json")
        if not os.path.exists(out_fn):
            todo.append((fn, out_fn))
    batch = max(1, int(os.environ.get("PREMVOS_DRIVER_BATCH", "1")))     # measured: host decode dominates, 1 is fastest
    from .. import io_pipeline as iop

    fast = isinstance(pred_func, OfflinePredictor) and not pred_func.net.mode_mask
    gpu_jpeg = fast and jpeg.enabled()            # entropy decode on the pool, inverse DCT / colour conversion on the GPU (BGR)

    def load(job):
        if gpu_jpeg:
            return job[1], jpeg.host_stage(job[0])
        img = np.asarray(Image.open(job[0]).convert("RGB"))[:, :, ::-1]             # cv2.imread gives BGR (train.py:500)
        return job[1], np.ascontiguousarray(img)

    def dump(out_fn, js):
        with open(out_fn, "w") as f:
            f.write(json.dumps(js))

    decoded = iop.prefetch(todo, load)            # JPEG decode runs ahead on a thread pool, JSON is written in the background
    n, held = 0, None
    with iop.Writer(enabled=iop.io_threads() > 0) as writer:
        while True:
            chunk = [held] if held is not None else []
            held = None
            for item in decoded:
                if chunk and (item[1].shape != chunk[0][1].shape or len(chunk) == batch):
                    held = item
                    break
                chunk.append(item)
            if not chunk:
                break
            orig = chunk[0][1].shape[:2]
            if fast:
                stage = _stage_for(pred_func.net, len(chunk))
                stage.run(jpeg.stack_frames([c[1] for c in chunk], stage.device, bgr=gpu_jpeg))   # (host arrays are BGR already)
                results = stage.json_results(orig)
            else:
                results = [convert_results_to_json(detect_one_image(c[1], pred_func)) for c in chunk]
            for (out_fn, _), res in zip(chunk, results):
                writer.submit(dump, out_fn, res)
                n += 1
    return n


def load_weights(path: str) -> Dict[str, object]:
    """``--load`` accepts what simple_run.sh:31-33 passes -- a TF checkpoint prefix (read without TensorFlow by
    premvos_amd.weights) -- or a torch pickle of the name->tensor dict documented in ProposalNet."""
    from ..weights import load_any
    return load_any(path, "proposal")


def infer_num_blocks(weights: Dict[str, object]):
    """(3,4,23,3) for the shipped ResNet-101; read off the variable names so reduced nets load too."""
    return tuple(len({k.split("/")[1] for k in weights if k.startswith(f"group{g}/block")}) for g in range(4))


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--load")
    ap.add_argument("--forward")
    ap.add_argument("--agnostic", action="store_true")
    ap.add_argument("--second_head", action="store_true")
    ap.add_argument("--forward_dataset", default="DAVIS")
    ap.add_argument("--davis_name")
    ap.add_argument("--generic_images_folder")
    ap.add_argument("--generic_images_pattern")
    a = ap.parse_args(argv)
    if not a.forward:
        raise SystemExit("only --forward (inference) is on the hot path; training is out of scope")
    if not a.agnostic:
        raise SystemExit("the shipped pipeline runs --agnostic (NUM_CLASS=2)")
    from .. import parallel
    parallel.bind_device()                      # one rank per GPU under torch.distributed.run
    w = load_weights(a.load)
    pred = OfflinePredictor(ProposalNet(w, num_blocks=infer_num_blocks(w)))
    forward(pred, a.forward, a.forward_dataset, a.davis_name, a.generic_images_folder, a.generic_images_pattern)
    return 0


if __name__ == "__main__":
    sys.exit(main())
