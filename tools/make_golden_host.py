"""Generates tests/golden/host_refs.json by IMPORTING / RUNNING the reference's pure-python host pieces that need neither
TensorFlow nor cv2 (run in the build container, where /root/reference exists; the fixture is data only):

  * proposal_net/utils/generate_anchors.py   generate_anchors() for the repo's config (stride 16, 5 sizes, 3 ratios)
  * proposal_net/utils/np_box_ops.py         area / intersection / iou / ioa on seeded boxes
  * proposal_net/combine_general_and_specific.py   run as a script on a small seeded tree of proposal JSON files
  * refinement_net/datasets/util/Normalization.py  normalize() on a seeded image
  * refinement_net/core/Config.py            typed getters + error behaviour on a synthetic config with that file's key types

Usage: python tools/make_golden_host.py [/root/reference]"""
import importlib.util
import json
import os
import runpy
import sys
import tempfile

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
CODE = os.path.join(REF, "code")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "host_refs.json")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def seeded_boxes(seed, n):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, 300, (n, 2))
    wh = rng.uniform(1, 200, (n, 2))
    return np.concatenate([xy, xy + wh], 1)          # np_box_ops takes [y_min, x_min, y_max, x_max]


def proposal_tree(root):
    """general has a/00000,a/00001,b/00000 ; specific has a/00001 (merged), a/00002 and c/00000 (specific-only),
    b/00000 is unreadable in specific (bare except -> [])."""
    rng = np.random.default_rng(5)

    def props(n, tag):
        return [{"bbox": [round(float(v), 1) for v in rng.uniform(0, 400, 4)], "score": round(float(rng.random()), 3),
                 "category_id": int(rng.integers(1, 80)), "tag": tag} for _ in range(n)]

    files = {"general_proposals/a/00000.json": props(3, "g"), "general_proposals/a/00001.json": props(2, "g"),
             "general_proposals/b/00000.json": props(1, "g"), "specific_proposals/a/00001.json": props(2, "s"),
             "specific_proposals/a/00002.json": props(1, "s"), "specific_proposals/c/00000.json": props(2, "s")}
    for rel, v in files.items():
        fn = os.path.join(root, "output", "intermediate", rel)
        os.makedirs(os.path.dirname(fn), exist_ok=True)
        with open(fn, "w") as f:
            json.dump(v, f)
    os.makedirs(os.path.join(root, "output", "intermediate", "specific_proposals", "b"), exist_ok=True)
    with open(os.path.join(root, "output", "intermediate", "specific_proposals", "b", "00000.json"), "w") as f:
        f.write("{not json")
    return files


def main():
    g = {}
    ga = _load(os.path.join(CODE, "proposal_net", "utils", "generate_anchors.py"), "ref_generate_anchors")
    sizes, ratios, stride = (32, 64, 128, 256, 512), (0.5, 1.0, 2.0), 16
    g["anchors"] = {"stride": stride, "sizes": sizes, "ratios": ratios,
                    "out": ga.generate_anchors(stride, scales=np.array(sizes, dtype=np.float64) / stride,
                                               ratios=np.array(ratios, dtype=np.float64)).tolist(),
                    "default": ga.generate_anchors().tolist()}

    nb = _load(os.path.join(CODE, "proposal_net", "utils", "np_box_ops.py"), "ref_np_box_ops")
    b1, b2 = seeded_boxes(1, 7), seeded_boxes(2, 5)
    g["np_box_ops"] = {"boxes1": b1.tolist(), "boxes2": b2.tolist(), "area": nb.area(b1).tolist(),
                       "intersection": nb.intersection(b1, b2).tolist(), "iou": nb.iou(b1, b2).tolist(),
                       "ioa": nb.ioa(b1, b2).tolist()}

    with tempfile.TemporaryDirectory() as td:
        files = proposal_tree(td)
        cwd = os.getcwd()
        os.chdir(td)
        try:
            runpy.run_path(os.path.join(CODE, "proposal_net", "combine_general_and_specific.py"), run_name="__main__")
        finally:
            os.chdir(cwd)
        comb = {}
        base = os.path.join(td, "output", "intermediate", "combined_proposals")
        for d, _, fs in os.walk(base):
            for fn in fs:
                with open(os.path.join(d, fn)) as f:
                    comb[os.path.relpath(os.path.join(d, fn), base)] = json.load(f)
        g["combine"] = {"inputs": files, "corrupt": "specific_proposals/b/00000.json", "combined": comb}

    nm = _load(os.path.join(CODE, "refinement_net", "datasets", "util", "Normalization.py"), "ref_normalization")
    img = np.random.default_rng(3).random((4, 5, 3)).astype(np.float32)
    g["normalize"] = {"img": img.tolist(), "out": nm.normalize(img.copy()).tolist(),
                      "mean": nm.IMAGENET_RGB_MEAN.tolist(), "std": nm.IMAGENET_RGB_STD.tolist()}

    cf = _load(os.path.join(CODE, "refinement_net", "core", "Config.py"), "ref_config")
    # a SYNTHETIC config with the key names / value types of refinement_net/configs/run (the fixture must hold data, not a copy
    # of one of the reference's files): '#' comment lines, strings, ints, floats, bools, lists, a nested dict, an int-key dict
    cfg_text = "\n".join([
        "# synthetic config for the Config-getter fixture",
        "{",
        '  "model": "synthetic_refiner", "load": "weights/some/prefix", "gpus": "0",',
        '  "image_input_dir": "data/frames", "model_dir": "models/",',
        "  # sizes",
        '  "batch_size": 3, "input_size_train": [385, 385], "bbox_jitter_factor": 0.05,',
        '  "use_bbox_guidance": true, "need_train": false,',
        '  "augmentors_train": ["bbox_jitter", "flip"],',
        '  "learning_rates": "{1: 1e-05, 4: 2.5e-06}",',
        '  "network": {"deeplab": {"class": "DeepLabV3Plus", "n_features": 2}, "output": {"class": "SegmentationSoftmax", "from": ["deeplab"]}}',
        "}", ""])
    with tempfile.TemporaryDirectory() as td:
        cfg_path = os.path.join(td, "run")
        with open(cfg_path, "w") as f:
            f.write(cfg_text)
        c = cf.Config(cfg_path)
    calls = [("string", "load", None), ("string", "model", None), ("dir", "image_input_dir", None),
             ("dir", "model_dir", None), ("int", "batch_size", None), ("int", "missing_int", 7),
             ("bool", "use_bbox_guidance", None), ("bool", "need_train", None), ("float", "bbox_jitter_factor", None),
             ("int_list", "input_size_train", None), ("string_list", "augmentors_train", None),
             ("int", "bbox_jitter_factor", None), ("string", "gpus", None), ("int", "missing_no_default", None),
             ("float", "gpus", None), ("bool", "gpus", None), ("int_key_dict", "learning_rates", None),
             ("dict", "network", None), ("string", "missing_str", "dflt"), ("int_list", "missing_list", [1, 2])]
    res = []
    for meth, key, default in calls:
        try:
            v = getattr(c, meth)(key, default)
            if meth == "int_key_dict":
                v = {str(k): x for k, x in v.items()}
            res.append({"method": meth, "key": key, "default": default, "value": v})
        except Exception as e:                                   # noqa: BLE001 -- the error type IS the vector
            res.append({"method": meth, "key": key, "default": default, "raises": type(e).__name__})
    g["config"] = {"text": cfg_text, "calls": res, "has": {k: c.has(k) for k in ("load", "nope")}}

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(g, f, indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
