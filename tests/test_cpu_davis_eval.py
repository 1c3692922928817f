"""tools/davis_eval.py (the J / F / J&F measures behind the reference's README.md:35-38 acceptance numbers) on synthetic masks
with hand-computable answers, and tools/accept_davis.py's input check naming what is missing."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_region_similarity_and_contour_accuracy_on_known_cases():
    E = _load("davis_eval")
    h, w = 100, 160
    a = np.zeros((h, w), bool)
    a[20:60, 30:90] = True                                   # 40 x 60 = 2400 pixels
    assert E.db_eval_iou(a, a) == 1.0 and E.db_eval_boundary(a, a) == 1.0
    empty = np.zeros_like(a)
    assert E.db_eval_iou(empty, empty) == 1.0 and E.db_eval_boundary(empty, empty) == 1.0
    assert E.db_eval_iou(a, empty) == 0.0 and E.db_eval_boundary(empty, a) == 0.0 and E.db_eval_boundary(a, empty) == 0.0
    b = np.roll(a, 10, axis=1)                               # shifted by 10 columns: intersection 40 x 50, union 40 x 70
    assert abs(E.db_eval_iou(a, b) - 2000.0 / 2800.0) < 1e-12
    # contour tolerance = ceil(0.008 * sqrt(100^2 + 160^2)) = ceil(1.509) = 2 pixels: a 1-pixel shift is a perfect contour, a 10-pixel
    # shift only matches the horizontal edges that overlap
    assert E.db_eval_boundary(np.roll(a, 1, axis=0), a) == 1.0
    f10 = E.db_eval_boundary(b, a)
    assert 0.3 < f10 < 0.7
    disjoint = np.zeros_like(a)
    disjoint[70:90, 100:150] = True
    assert E.db_eval_iou(a, disjoint) == 0.0 and E.db_eval_boundary(disjoint, a) == 0.0
    # boundary map: every mask pixel with a differing east / south / south-east neighbour
    sq = np.zeros((6, 6), bool)
    sq[2:4, 2:4] = True
    assert E.seg2bmap(sq).sum() == 8 and E.seg2bmap(sq)[1, 1] and not E.seg2bmap(sq)[2, 2] and E.seg2bmap(sq)[3, 3]


def test_sequence_protocol_first_and_last_frame_excluded_objects_by_id(tmp_path):
    from PIL import Image
    E = _load("davis_eval")
    h, w, T = 60, 80, 5

    def save(arr, fn):                                        # a palette PNG whose pixel values are the object ids (DAVIS / save_pngs)
        im = Image.frombytes("P", (arr.shape[1], arr.shape[0]), arr.tobytes())
        im.putpalette([0, 0, 0, 128, 0, 0, 0, 128, 0] + [0] * (3 * 253))
        im.save(fn)
    ann, res = tmp_path / "ann" / "seq", tmp_path / "res" / "seq"
    ann.mkdir(parents=True)
    res.mkdir(parents=True)
    for t in range(T):
        g = np.zeros((h, w), np.uint8)
        g[10:30, 10 + t:40 + t] = 1
        g[35:55, 30:70] = 2
        save(g, ann / f"{t:05d}.png")
        r = g.copy()
        if t in (0, T - 1):
            r[:] = 0                                          # garbage on the excluded frames must not matter
        if t == 2:
            r[r == 2] = 0                                     # object 2 lost in one of the three evaluated frames
        save(r, res / f"{t:05d}.png")
    out = E.evaluate(str(tmp_path / "res"), str(tmp_path / "ann"))
    assert out["objects"] == 2 and out["sequences"] == 1
    o = out["per_sequence"]["seq"]
    assert o["1"] == {"J": 1.0, "F": 1.0}
    assert abs(o["2"]["J"] - 2.0 / 3.0) < 1e-5 and abs(o["2"]["F"] - 2.0 / 3.0) < 1e-5
    assert abs(out["mean_JF_percent"] - 100.0 * (1.0 + 2.0 / 3.0) / 2.0) < 1e-3
    # a missing result frame counts as an empty mask
    os.remove(res / "00001.png")
    out2 = E.evaluate(str(tmp_path / "res"), str(tmp_path / "ann"))
    assert abs(out2["per_sequence"]["seq"]["1"]["J"] - 2.0 / 3.0) < 1e-5


def test_accept_davis_names_what_is_missing(tmp_path, capsys):
    A = _load("accept_davis")
    (tmp_path / "seq_to_run.txt").write_text("data/DAVIS/JPEGImages/480p/bear/\n")
    assert A.main(["--root", str(tmp_path), "--check-only", "--skip-reid"]) == 2
    msg = capsys.readouterr().out
    assert "bear" in msg and "pwc_net.pth.tar" in msg and "proposal_general_weights" in msg and "refinement_specific_weights" in msg
    # evaluation-only mode without MergeTrack's output explains how to get it
    assert A.main(["--root", str(tmp_path), "--skip-stages"]) == 2
    assert "MergeTrack/merge.py" in capsys.readouterr().out
