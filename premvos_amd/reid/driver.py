"""ReID stage driver: the counterpart of ``ReID_net/main.py configs/run`` (batch stage) and of the in-process API
MergeTrack uses (MergeTrack/ReID_net_functions.py:19-45).

Reference boundary kept:
  * batch stage (datasets/Similarity/DAVIS_Forward_Similarity.py:25-43 + Forwarding/ReIDForwarding.py:34-92): every
    <bb_input_dir>/<seq>/<frame>.json is copied to <output_dir>/<seq>/<frame>.json with "ReID": [128 floats] added to
    each proposal whose ``toBbox(segmentation)`` has w > 0 and h > 0 (the others stay without the key);
  * in-process: ``engine = ReID_net_init()``; ``add_ReID(proposals, image_fn, engine)`` -> every proposal's 'bbox' (xywh)
    gets its embedding (crops of the in-merge feed dataset: excess >= 1 pixel, zero image for boxes <= 10 px).
"""
from __future__ import annotations

import glob
import json
import os
import sys
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import jpeg, rle
from ..refinement.driver import Config as _TypedConfig
from .model import ReIDNet

BATCH = 40


class Config(_TypedConfig):
    """ReID_net/Config.py spells the string getter ``str`` (configs are read with ``config.str("model")``)."""

    def str(self, key, default=None):
        return self.string(key, default)


def units_from_config(network: Dict[str, dict]):
    """The ``"network"`` table of the config (configs/run:37-70) -> the (name, n_features, filter sizes, strides) list
    ``ReIDNet`` takes, with ResidualUnit2's defaults (NetworkLayers.py:148-166): 2 convs, 3x3 filters, stride 1,
    n_features = the input's.  Only the layer classes of the shipped ReID net are accepted."""
    units, feats_in = [], None
    order = []
    for name, spec in network.items():
        cls = spec["class"]
        if cls == "Conv" and "from" not in spec:
            assert spec.get("activation") == "linear" and not spec.get("batch_norm", False), "unsupported stem"
            feats_in = spec["n_features"]
        elif cls == "ResidualUnit2":
            assert feats_in is not None and spec["from"] == [order[-1]], "the ReID net is a plain chain"
            n = spec.get("n_convs", 2)
            f = spec.get("n_features", feats_in)
            f = list(f) if isinstance(f, list) else [f] * n
            ks = [k[0] for k in spec.get("filter_size", [[3, 3]] * n)]
            st = [s[0] for s in spec.get("strides", [[1, 1]] * n)]
            assert len(f) == len(ks) == len(st) == n and "dilations" not in spec
            units.append((name, tuple(f), tuple(ks), tuple(st)))
            feats_in = f[-1]
        elif cls in ("Conv", "FullyConnected", "FullyConnectedWithTripletLoss"):
            pass                                   # conv1 / fc1 / fc2 / outputTriplet: shapes come from the weights
        else:
            raise ValueError(f"unsupported layer class {cls!r} in the ReID network table")
        order.append(name)
    return units


def load_weights(path: str) -> Dict[str, object]:
    from ..weights import load_any
    return load_any(path, "reid")


class ReIDEngine:
    def __init__(self, net: ReIDNet, max_boxes: int = BATCH):
        self.net, self.max_boxes = net, max_boxes

    def embed(self, image_rgb: np.ndarray, boxes_xywh, feed: bool) -> np.ndarray:
        boxes = np.asarray(boxes_xywh, np.float32).reshape(-1, 4)
        out = np.zeros((len(boxes), 128), np.float32)
        frame = jpeg.to_device(image_rgb, self.net.device)
        for s in range(0, len(boxes), self.max_boxes):
            chunk = boxes[s:s + self.max_boxes]
            P = self.max_boxes if len(boxes) > self.max_boxes else _bucket(len(chunk))
            out[s:s + len(chunk)] = self.net.embed(frame, chunk, max_boxes=P, feed=feed).cpu().numpy()
        return out


def _bucket(n: int) -> int:
    for b in (1, 2, 4, 8, 12, 20, 40):
        if n <= b:
            return b
    return n


def engine_from_config(cfg: "Config") -> ReIDEngine:
    units = units_from_config(cfg.dict("network")) if cfg.has("network") else None
    return ReIDEngine(ReIDNet(load_weights(cfg.str("load")), **({"units": units} if units else {})))


def ReID_net_init(config_path: str = "ReID_net/configs/live") -> ReIDEngine:
    return engine_from_config(Config(config_path))


def add_ReID(proposals: List[dict], image_fn: str, ReID_net: ReIDEngine) -> List[dict]:
    """MergeTrack/ReID_net_functions.py:26-45."""
    from PIL import Image
    if not proposals:
        return proposals
    image = np.asarray(Image.open(image_fn).convert("RGB"))
    emb = ReID_net.embed(image, [p["bbox"] for p in proposals], feed=True)
    for p, e in zip(proposals, emb):
        p["ReID"] = e.tolist()
    return proposals


def forward_directory(engine: ReIDEngine, image_input_dir: str, bb_input_dir: str, output_dir: str) -> int:
    """Forwarding/ReIDForwarding.py:33-92: every refined proposal with a non-empty mask gains its 128-d embedding.  JSON
    parsing, RLE -> bbox and JPEG decode run ahead on a thread pool, JSON writing on a background thread
    (premvos_amd.io_pipeline); the file contents equal the serial loop's."""
    from PIL import Image
    from .. import io_pipeline as iop
    from .. import sidecar as sc
    # <frame>.json (the reference's format) or, from this package's refinement stage with PREMVOS_SIDECAR=1, <frame>.pmv (bit-packed
    # masks: no RLE string to parse); the output keeps the input's format
    files = sorted(glob.glob(os.path.join(bb_input_dir, "*", "*.json")) + glob.glob(os.path.join(bb_input_dir, "*", "*" + sc.EXT)))
    read_frame = jpeg.loader()                    # PIL on the host, or (PREMVOS_GPU_JPEG=1) entropy decode here + the rest on the GPU

    def load(jf):
        rel = os.path.relpath(jf, bb_input_dir)
        if jf.endswith(sc.EXT):
            d = sc.read(jf)
            proposals = d
            bbs = [sc.tight_bbox(m) for m in sc.unpack_masks(d)] if d["mask_bits"] is not None else []
        else:
            with open(jf) as f:
                proposals = json.load(f)
            bbs = [rle.to_bbox(p["segmentation"]) for p in proposals]
        boxes, idx = [], []
        for i, bb in enumerate(bbs):
            if bb[2] <= 0 or bb[3] <= 0:
                continue
            boxes.append(bb)
            idx.append(i)
        image = None
        if boxes:
            image = read_frame(os.path.join(image_input_dir, os.path.splitext(rel)[0] + ".jpg"))
        return rel, proposals, boxes, idx, image

    def dump(out_fn, proposals):
        os.makedirs(os.path.dirname(out_fn), exist_ok=True)
        if isinstance(proposals, dict):                      # side-car in, side-car out
            sc.write_dict(out_fn, proposals)
            return
        with open(out_fn, "w") as f:
            f.write(json.dumps(proposals))     # (same text as json.dump, the encoder's C core in one shot)

    n = 0
    with iop.Writer(enabled=iop.io_threads() > 0) as writer:
        for rel, proposals, boxes, idx, image in iop.prefetch(files, load):
            if boxes:
                emb = engine.embed(image, boxes, feed=False)
                if isinstance(proposals, dict):
                    n_p = len(proposals["bbox"])
                    proposals["reid"], proposals["has_reid"] = np.zeros((n_p, 128), np.float32), np.zeros(n_p, np.uint8)
                    for i, e in zip(idx, emb):
                        proposals["reid"][i], proposals["has_reid"][i] = np.asarray(e, np.float32), 1
                else:
                    for i, e in zip(idx, emb):
                        proposals[i]["ReID"] = np.array(e).tolist()
            writer.submit(dump, os.path.join(output_dir, rel), proposals)
            n += 1
    return n


def main(argv: Optional[List[str]] = None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    assert len(argv) == 1, "usage: driver.py <config>"
    cfg = Config(argv[0])
    engine = engine_from_config(cfg)
    forward_directory(engine, cfg.dir("image_input_dir"), cfg.dir("bb_input_dir"), cfg.dir("output_dir"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
