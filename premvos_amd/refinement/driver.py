"""Refinement stage driver: the counterpart of refinement_net/main.py + the few-shot forwarder, and of the
in-process API MergeTrack uses.

Reference boundary kept:
  * JSON config with '#' comment lines and typed getters (core/Config.py:5-91); keys used: image_input_dir,
    bb_input_dir, output_dir, load, input_size_train, use_bbox_guidance  (configs/run:2-37)
  * batch stage: for every frame JSON under bb_input_dir/<seq>/<frame>.json write output_dir/<seq>/<frame>.json
    = same list, same order, each proposal gaining "segmentation" {"size":[H,W],"counts":str} and "conf_score"
    (a *string*)                                           (forwarding/FewShotSegmentationForwarder.py:85-155)
  * in-process: ``engine = refinement_net_init()``; ``do_refinement(proposals, image_fn, engine)`` mutates and
    returns the proposal list                              (MergeTrack/refinement_net_functions.py:19-24,38-64)
"""
from __future__ import annotations

import glob
import json
from collections import OrderedDict
import os
import sys
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import jpeg, rle
from .model import RefinementNet


class Config:
    """The refinement_net config file: a JSON object in which lines starting with '#' are comments, optionally overlaid
    by a second JSON object (main.py's update string), read through typed getters.

    Interface contract taken from refinement_net/core/Config.py:5-91 and pinned by tests/golden/host_refs.json (generated
    by running that class): getters never coerce -- a stored value of another JSON type raises TypeError (bool is an int
    for Python, so ``int("flag")`` accepts it, as there); a missing key needs a default (AssertionError otherwise); a
    default of the wrong type is an AssertionError even when the key exists; list getters check every element.
    """

    _SCALARS = {"bool": bool, "string": str, "int": int, "float": float, "dict": dict}
    _LISTS = {"int_list": int, "float_list": float, "string_list": str}

    def __init__(self, filename: str, update_config_string: str = ""):
        with open(filename) as f:
            text = "".join("\n" if ln.lstrip().startswith("#") else ln for ln in f)
        self._entries = json.loads(text, object_pairs_hook=OrderedDict)
        if update_config_string:
            self._entries.update(json.loads(update_config_string, object_pairs_hook=OrderedDict))

    def has(self, key) -> bool:
        return key in self._entries

    def _get(self, key, elem_type, default, as_list: bool):
        """One getter for all types: ``as_list`` -> a list whose items are ``elem_type``; else a single ``elem_type``."""
        def conforms(v):
            return (isinstance(v, list) and all(isinstance(x, elem_type) for x in v)) if as_list else isinstance(v, elem_type)
        assert default is None or conforms(default)
        if key not in self._entries:
            assert default is not None
            return default
        v = self._entries[key]
        if as_list:
            assert conforms(v)
        elif not conforms(v):
            raise TypeError()
        return v

    def __getattr__(self, name):
        # bool / string / int / float / dict / int_list / float_list / string_list (key, default=None)
        if name in Config._SCALARS:
            t = Config._SCALARS[name]
            return lambda key, default=None: self._get(key, t, default, False)
        if name in Config._LISTS:
            t = Config._LISTS[name]
            return lambda key, default=None: self._get(key, t, default, True)
        raise AttributeError(name)

    def int_key_dict(self, key, default=None):
        """A dict with integer keys, stored as the *string* of a Python literal (JSON keys cannot be ints)."""
        import ast
        text = self._get(key, str, "", False)
        out = ast.literal_eval(text) if text else default
        assert out is not None
        assert isinstance(out, dict) and all(isinstance(k, int) for k in out)
        assert default is None or (isinstance(default, dict) and all(isinstance(k, int) for k in default))
        return out

    def dir(self, key, default=None) -> str:
        """A directory path, always with a trailing '/'."""
        path = self._get(key, str, default, False)
        return path if path.endswith("/") else path + "/"


def _boxes_from_proposals(proposals: List[dict]) -> np.ndarray:
    """xywh JSON box -> [y0,x0,y1,x1] (DAVISFewShotSegmentationDataset.py:55-59)."""
    out = np.zeros((len(proposals), 4), np.float32)
    for i, p in enumerate(proposals):
        x0, y0, w, h = p["bbox"]
        out[i] = [y0, x0, y0 + h, x0 + w]
    return out


class Extractions:
    """Key names of refinement_net/core/Extractions.py:1-6."""
    EXTRACTIONS = "extractions"
    SEGMENTATION_POSTERIORS = "segmentation_posteriors"
    SEGMENTATION_POSTERIORS_ORIGINAL_SIZE = "segmentation_posteriors_original_size"
    SEGMENTATION_MASK_ORIGINAL_SIZE = "segmentation_mask_original_size"
    SEGMENTATION_MASK_INPUT_SIZE = "segmentation_mask_input_size"


class DataKeys:
    """The keys of refinement_net/datasets/DataKeys.py the feed dataset uses."""
    IMAGES = "images"
    SEGMENTATION_LABELS = "segmentation_labels"
    IMAGE_FILENAMES = "image_filenames"
    BBOXES_y0x0y1x1 = "bboxes_y0x0y1x1"
    OBJ_TAGS = "obj_tags"


MEASURES = "measures"          # core/Measures.py


class _FeedData:
    """``engine.valid_data`` of the reference (FewShotFeedSegmentationDataset.py:24-51): the per-image example table and
    the per-box "feed dict".  Here a feed dict just names (table, index); the table additionally remembers the uint8
    frame so the engine can refine ALL its boxes in one batched GPU pass the first time any of them is asked for."""

    def set_up_data_for_image(self, image, boxes):
        """The reference's example table for one frame (FewShotFeedSegmentationDataset.py:35-51): {box index: example} with the
        frame scaled to [0, 1], an all-zero label plane, the box converted from x, y, w, h to y0, x0, y1, x1 and the index as
        the object tag -- or None when the frame has no boxes."""
        if len(boxes) == 0:
            return None
        pixels = np.asarray(image)
        scaled, blank = pixels / 255, np.zeros(pixels.shape[:2] + (1,), dtype=np.uint8)

        def example(k, xywh):
            x, y, w, h = xywh
            return {DataKeys.IMAGES: scaled, DataKeys.SEGMENTATION_LABELS: blank, DataKeys.IMAGE_FILENAMES: "",
                    DataKeys.BBOXES_y0x0y1x1: [y, x, y + h, x + w], DataKeys.OBJ_TAGS: str(k)}
        table = {k: example(k, b) for k, b in enumerate(boxes)}
        # remember the uint8 frame of the (at most two) most recent tables: the first box asked for refines them all at once
        self._frames[id(table)] = (table, np.ascontiguousarray(pixels[:, :, :3], dtype=np.uint8))
        while len(self._frames) > 2:
            self._frames.pop(next(iter(self._frames)))
        return table

    def __init__(self):
        self._frames: "OrderedDict[int, tuple]" = OrderedDict()

    def get_feed_dict_for_next_step(self, image_data, bbox_idx):
        assert bbox_idx in image_data, bbox_idx
        return {"image_data": image_data, "bbox_idx": bbox_idx}


class _FeedTrainer:
    """``engine.trainer`` of the reference: ``validation_step(feed_dict=..., extraction_keys=[...])`` ->
    {"measures": {}, "extractions": {key: [array[1, ...]]}}   (core/Trainer.py:128-169, Extractions.py:9-23)."""

    def __init__(self, engine: "RefinementEngine"):
        self._engine = engine
        self._cache = (None, None)
        self.validation_step_number = 0

    def validation_step(self, epoch=None, feed_dict=None, extraction_keys=()):
        image_data, idx = feed_dict["image_data"], feed_dict["bbox_idx"]
        if self._cache[0] is not image_data:
            entry = self._engine.valid_data._frames.get(id(image_data))
            if entry is not None and entry[0] is image_data:
                frame = entry[1]
            else:                                                          # table built by someone else: recover the frame
                frame = np.rint(np.asarray(image_data[idx][DataKeys.IMAGES]) * 255).astype(np.uint8)
            boxes = np.array([image_data[i][DataKeys.BBOXES_y0x0y1x1] for i in range(len(image_data))], np.float32)
            self._cache = (image_data, self._engine.refine_boxes(frame, boxes))
        masks, post, _ = self._cache[1]
        self.validation_step_number += 1
        ex = {Extractions.SEGMENTATION_MASK_ORIGINAL_SIZE: [masks[idx:idx + 1].astype(np.int64)],
              Extractions.SEGMENTATION_POSTERIORS_ORIGINAL_SIZE: [post[idx:idx + 1]],
              DataKeys.OBJ_TAGS: [np.array([image_data[idx][DataKeys.OBJ_TAGS].encode("utf-8")])]}
        return {MEASURES: {}, Extractions.EXTRACTIONS: {k: v for k, v in ex.items() if k in extraction_keys}}


class RefinementEngine:
    """What MergeTrack holds as ``engine``: refines all proposals of a frame in one batched pass.  ``valid_data`` and
    ``trainer`` reproduce the reference engine's call shapes, so MergeTrack/refinement_net_functions.py:38-64 runs
    against this object unchanged (its per-box loop then reads results of ONE batched pass per image)."""

    def __init__(self, net: RefinementNet, max_boxes: int = 40):
        self.net, self.max_boxes = net, max_boxes
        self.valid_data = _FeedData()
        self.trainer = _FeedTrainer(self)

    def refine_frames(self, images, proposal_lists, lane: int = 0, sidecar: bool = False, defer: bool = False):
        """Several frames of equal size at once: the crops of all of them form one batch, PACKED (``RefinementNet.refine_packed``:
        sum(n_i) slots rounded up to the next bucket, not frames x max(n_i)).
        ``lane`` selects an independent workspace of the net, so calls on different lanes may run concurrently (each on the
        calling thread's current stream).  ``sidecar``: instead of "segmentation" (COCO RLE) / "conf_score" (str) the proposals get
        "mask_bits" (packed on the GPU, premvos_amd.sidecar layout) and "conf" (float32) -- the optional binary fast path.
        ``defer``: the GPU work and the device-to-host copies happen here, the host-side packing of the RLE strings is returned
        as a callable (run it on another thread, e.g. the file writer's: it fills the proposal dicts); None when nothing is left."""
        live = [(im, pr) for im, pr in zip(images, proposal_lists) if pr]
        if not live:
            return None if defer else proposal_lists
        if len(live) == 1 or max(len(pr) for _, pr in live) > self.max_boxes:
            for im, pr in live:
                self.refine_frame(im, pr, lane=lane, sidecar=sidecar)
            return None if defer else proposal_lists
        total = sum(len(pr) for _, pr in live)
        boxes = [_boxes_from_proposals(pr) for _, pr in live]
        frames = jpeg.stack_frames([im for im, _ in live], self.net.device)
        group = max(len(live), int(os.environ.get("PREMVOS_DRIVER_BATCH", "4")))
        p = self.net.refine_packed(frames, [torch.from_numpy(b) for b in boxes], _bucket_total(total), group, lane=lane)
        sel = p.mask_g[0, :total]                                                          # the slots are contiguous: no gather
        conf = p.conf_g[0, :total].cpu().numpy()
        flat = [pr[i] for _, pr in live for i in range(len(pr))]
        if sidecar:
            segs = _pack_on_gpu(sel)
            for k, q in enumerate(flat):
                q["mask_bits"], q["conf"] = segs[k], conf[k]
            return None if defer else proposal_lists
        from ..mergetrack import encode_masks_begin, encode_masks_finish
        handle = encode_masks_begin(sel)                                                   # run boundaries on the GPU + their D2H copy

        def finish():
            segs = encode_masks_finish(handle)                                             # differencing + ASCII packing (host only)
            for k, q in enumerate(flat):
                q["segmentation"] = segs[k]
                q["conf_score"] = str(conf[k])
        if defer:
            return finish
        finish()
        return proposal_lists

    def refine_frames_device(self, images, proposal_lists, out_masks: torch.Tensor, out_conf: torch.Tensor, lane: int = 0) -> None:
        """``refine_frames`` for a caller that keeps the results in HBM (premvos_amd.stream --gather): the masks of frame j's n_j
        proposals go to ``out_masks[j, :n_j, :H, :W]`` (uint8, a block at least as large as the frame), their conf to
        ``out_conf[j, :n_j]``; nothing crosses PCIe, no RLE string is built.  Same packed launches as ``refine_frames``."""
        live = [(j, im, pr) for j, (im, pr) in enumerate(zip(images, proposal_lists)) if pr]
        if not live:
            return
        H, W = live[0][1].shape[:2]
        if len(live) == 1 or max(len(pr) for _, _, pr in live) > self.max_boxes:
            for j, im, pr in live:
                boxes = _boxes_from_proposals(pr)
                frame = jpeg.to_device(im, self.net.device)
                for s0 in range(0, len(pr), self.max_boxes):
                    chunk = boxes[s0:s0 + self.max_boxes]
                    P = self.max_boxes if len(pr) > self.max_boxes else _bucket(len(chunk))
                    p = self.net.refine(frame, torch.from_numpy(chunk).to(self.net.device), max_boxes=P, lane=lane)
                    out_masks[j, s0:s0 + len(chunk), :H, :W].copy_(p.mask[:len(chunk)])
                    out_conf[j, s0:s0 + len(chunk)].copy_(p.conf[:len(chunk)])
            return
        total = sum(len(pr) for _, _, pr in live)
        boxes = [_boxes_from_proposals(pr) for _, _, pr in live]
        frames = jpeg.stack_frames([im for _, im, _ in live], self.net.device)
        group = max(len(live), int(os.environ.get("PREMVOS_DRIVER_BATCH", "4")))
        p = self.net.refine_packed(frames, [torch.from_numpy(b) for b in boxes], _bucket_total(total), group, lane=lane)
        off = 0
        for j, _, pr in live:                       # the slots of a frame are contiguous
            n = len(pr)
            out_masks[j, :n, :H, :W].copy_(p.mask_g[0, off:off + n])
            out_conf[j, :n].copy_(p.conf_g[0, off:off + n])
            off += n

    def refine_boxes(self, frame_u8: np.ndarray, boxes_y0x0y1x1: np.ndarray):
        """-> (mask uint8 [n,H,W], posterior f32 [n,H,W], conf f32 [n]) as numpy."""
        n = len(boxes_y0x0y1x1)
        H, W = frame_u8.shape[:2]
        masks = np.zeros((n, H, W), np.uint8)
        post = np.zeros((n, H, W), np.float32)
        conf = np.zeros((n,), np.float32)
        frame = torch.from_numpy(np.ascontiguousarray(frame_u8[:, :, :3])).to(self.net.device)
        for s in range(0, n, self.max_boxes):
            chunk = np.asarray(boxes_y0x0y1x1[s:s + self.max_boxes], np.float32)
            P = self.max_boxes if n > self.max_boxes else _bucket(len(chunk))
            p = self.net.refine(frame, torch.from_numpy(chunk).to(self.net.device), max_boxes=P, with_posterior=True)
            masks[s:s + len(chunk)] = p.mask[:len(chunk)].cpu().numpy()
            post[s:s + len(chunk)] = p.posterior[:len(chunk)].cpu().numpy()
            conf[s:s + len(chunk)] = p.conf[:len(chunk)].cpu().numpy()
        return masks, post, conf

    def refine_frame(self, image_rgb: np.ndarray, proposals: List[dict], lane: int = 0, sidecar: bool = False) -> List[dict]:
        if not proposals:
            return proposals
        boxes = _boxes_from_proposals(proposals)
        frame = jpeg.to_device(image_rgb, self.net.device)
        for s in range(0, len(proposals), self.max_boxes):
            chunk = boxes[s:s + self.max_boxes]
            P = self.max_boxes if len(proposals) > self.max_boxes else _bucket(len(chunk))
            p = self.net.refine(frame, torch.from_numpy(chunk).to(self.net.device), max_boxes=P, lane=lane)
            conf = p.conf[:len(chunk)].cpu().numpy()
            segs = _pack_on_gpu(p.mask[:len(chunk)]) if sidecar else _encode_on_gpu(p.mask[:len(chunk)])
            for i in range(len(chunk)):
                if sidecar:
                    proposals[s + i]["mask_bits"], proposals[s + i]["conf"] = segs[i], conf[i]
                else:
                    proposals[s + i]["segmentation"] = segs[i]
                    proposals[s + i]["conf_score"] = str(conf[i])
        return proposals


def _encode_on_gpu(masks: torch.Tensor) -> List[dict]:
    """COCO RLE of uint8 CUDA masks [n,H,W] (same strings as rle.encode; premvos_rle_boundaries_u8 + host differencing)."""
    from ..mergetrack import encode_masks
    return encode_masks(masks)


def _pack_on_gpu(masks: torch.Tensor) -> np.ndarray:
    """uint8 CUDA masks [n,H,W] -> packed bits [n, ceil(H*W/8)] on the host (premvos_mask_pack_bits_u8; one launch when a mask
    is a whole number of bytes, else one per mask)."""
    from .. import _lib
    n, h, w = masks.shape
    nbytes = (h * w + 7) // 8
    m = masks.contiguous()
    bits = torch.empty((n, nbytes), dtype=torch.uint8, device=m.device)
    lib = _lib.load()
    if (h * w) % 8 == 0:
        _lib.check(lib.premvos_mask_pack_bits_u8(m.data_ptr(), n * h * w, bits.data_ptr(), _lib.current_stream()), "mask_pack_bits")
    else:
        for i in range(n):
            _lib.check(lib.premvos_mask_pack_bits_u8(m[i].data_ptr(), h * w, bits[i].data_ptr(), _lib.current_stream()), "mask_pack_bits")
    return bits.cpu().numpy()


def _bucket(n: int) -> int:
    """Boxes per frame of the launch plan: padded slots cost full network FLOPs (a frame with 22 combined proposals in a
    40-box plan runs 1.8x the work), so the steps are fine; every distinct size is its own plan (LRU-bounded in the net)."""
    for b in (1, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 32, 36, 40):
        if n <= b:
            return b
    return n


def _bucket_total(n: int) -> int:
    """Slots of a packed group plan: the per-frame steps up to 40, then steps of 8 (one plan -- a full activation set -- per size)."""
    return _bucket(n) if n <= 40 else (n + 7) // 8 * 8


def do_refinement(proposals: List[dict], image_fn: str, refinement_net: RefinementEngine) -> List[dict]:
    """MergeTrack/refinement_net_functions.py:38-64."""
    from PIL import Image
    image = np.asarray(Image.open(image_fn).convert("RGB"))
    return refinement_net.refine_frame(image, proposals)


def load_weights(path: str) -> Dict[str, object]:
    """config key "load" (configs/run:9): a TF checkpoint prefix (read without TensorFlow) or a torch pickle."""
    from ..weights import load_any
    return load_any(path, "refinement")


def infer_num_middle(weights: Dict[str, object]) -> int:
    return len({k.split("/")[2] for k in weights if k.startswith("middle_flow/block1/unit_")})


def refinement_net_init(config_path: str = "refinement_net/configs/live") -> RefinementEngine:
    cfg = Config(config_path)
    w = load_weights(cfg.string("load"))
    return RefinementEngine(RefinementNet(w, infer_num_middle(w)))


def forward_directory(engine: RefinementEngine, image_input_dir: str, bb_input_dir: str, output_dir: str,
                      sidecar: Optional[bool] = None) -> int:
    """The batch stage: every <seq>/<frame>.json of bb_input_dir -> output_dir (same relative name).  Consecutive frames of
    equal size are refined as one batch of crops (every box is an independent example, FewShotSegmentationForwarder.py:104-110).
    Host work overlaps the GPU (premvos_amd.io_pipeline): JPEG decode + JSON parsing run ahead on a thread pool, groups
    alternate over two workspace lanes of the net so one group's RLE packing overlaps the next group's kernels, and the JSON
    files are written by a background thread -- same bytes as the serial driver (PREMVOS_IO_THREADS=0 PREMVOS_IO_LANES=1)."""
    from PIL import Image
    from .. import io_pipeline as iop
    group = max(1, int(os.environ.get("PREMVOS_DRIVER_BATCH", "4")))
    files = sorted(glob.glob(os.path.join(bb_input_dir, "*", "*.json")))
    from .. import parallel
    if parallel.env_rank()[0] > 1:              # under torch.distributed.run: this rank's slice of the videos (the reference's
        mine = set(parallel.my_videos(sorted({os.path.basename(os.path.dirname(f)) for f in files})))   # curr_run_num scheme)
        files = [f for f in files if os.path.basename(os.path.dirname(f)) in mine]
    # optional binary fast path: <frame>.pmv with bit-packed masks instead of <frame>.json with RLE strings;
    # read by this package's ReID stage, `python -m premvos_amd.sidecar --to-json` gives MergeTrack its JSON back
    sidecar = os.environ.get("PREMVOS_SIDECAR", "0") == "1" if sidecar is None else sidecar
    read_frame = jpeg.loader()                    # PIL on the host, or (PREMVOS_GPU_JPEG=1) entropy decode here + the rest on the GPU

    def load(jf):
        rel = os.path.relpath(jf, bb_input_dir)
        with open(jf) as f:
            proposals = json.load(f)
        image = read_frame(os.path.join(image_input_dir, os.path.splitext(rel)[0] + ".jpg"))
        return os.path.join(output_dir, rel), image, proposals

    def groups():
        jobs = []
        for job in iop.prefetch(files, load):
            if jobs and (len(jobs) == group or jobs[0][1].shape != job[1].shape):
                yield jobs
                jobs = []
            jobs.append(job)
        if jobs:
            yield jobs

    n_lanes = iop.io_lanes()
    streams = [torch.cuda.Stream(device=engine.net.device) for _ in range(n_lanes)] if n_lanes > 1 else [None]
    if n_lanes > 1:
        engine.net.use_graph = False      # lanes launch eagerly: no HIP-graph capture on one host thread while another launches

    def work(lane, jobs):                   # GPU half here; the RLE strings are packed by the writer thread (``finish``)
        if streams[lane] is None:
            finish = engine.refine_frames([j[1] for j in jobs], [j[2] for j in jobs], sidecar=sidecar, defer=True)
        else:
            with torch.cuda.stream(streams[lane]):
                finish = engine.refine_frames([j[1] for j in jobs], [j[2] for j in jobs], lane=lane, sidecar=sidecar, defer=True)
                streams[lane].synchronize()
        return jobs, finish

    def write(jobs, finish):
        if finish is not None:
            finish()
        (_write_sidecars if sidecar else _write_jobs)(jobs)

    with iop.Writer(enabled=iop.io_threads() > 0) as writer:
        for jobs, finish in iop.lanes(groups(), work, n_lanes):
            writer.submit(write, jobs, finish)
    return len(files)


def _write_sidecars(jobs) -> None:
    from .. import sidecar as sc
    for out_fn, image, proposals in jobs:
        os.makedirs(os.path.dirname(out_fn), exist_ok=True)
        h, w = image.shape[:2]
        n = len(proposals)
        sc.write(os.path.splitext(out_fn)[0] + sc.EXT, h, w, np.array([p["bbox"] for p in proposals], np.float64).reshape(n, 4),
                 np.array([p["score"] for p in proposals], np.float64), np.array([p["conf"] for p in proposals], np.float32),
                 np.stack([p["mask_bits"] for p in proposals]) if n else np.zeros((0, (h * w + 7) // 8), np.uint8))


def _write_jobs(jobs) -> None:
    for out_fn, _, proposals in jobs:
        os.makedirs(os.path.dirname(out_fn), exist_ok=True)
        with open(out_fn, "w") as f:
            f.write(json.dumps(proposals))     # (same text as json.dump, the encoder's C core in one shot)


def main(argv: Optional[List[str]] = None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    assert len(argv) in (1, 2), "usage: driver.py <config> [update_config_string]"
    cfg = Config(argv[0], argv[1] if len(argv) > 1 else "")
    from .. import parallel
    parallel.bind_device()                      # one rank per GPU under torch.distributed.run
    w = load_weights(cfg.string("load"))
    engine = RefinementEngine(RefinementNet(w, infer_num_middle(w)))
    forward_directory(engine, cfg.dir("image_input_dir"), cfg.dir("bb_input_dir"), cfg.dir("output_dir"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
