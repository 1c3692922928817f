"""premvos_write_frame_files_host (csrc/host_files.hip): the interpreter-free writer of a frame's files against the interpreter.

The merge rank of a gathered multi-GPU job writes every rank's files; the Python writers held the interpreter lock for two thirds of
a frame's host work.  The C twin must produce THEIR bytes: Python's repr(float) (json.dump), numpy's str(float32) (conf_score),
numpy's float32 arithmetic of eval.py:93-94 / train.py:388-428 (divide by the scale, clip, xywh, round), COCO count strings with
their one JSON escape, .flo layout."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from premvos_amd import _lib, rle
from premvos_amd.flow.driver import flo_bytes
from premvos_amd.proposal.driver import results_json


def _fmt(vals, f32):
    lib = _lib.load()
    v = np.ascontiguousarray(vals, dtype=np.float64)
    buf = C.create_string_buffer(40 * len(v) + 16)
    n = lib.premvos_format_floats_host(v.ctypes.data, len(v), int(f32), buf, len(buf))
    assert n >= 0, lib.premvos_last_error()
    return buf.raw[:n].decode().split("\n")[:-1]


def test_double_repr_is_pythons_on_widened_float32_decimals_and_general_doubles():
    rng = np.random.default_rng(0)
    a = np.round(rng.uniform(-50, 2000, 40000).astype(np.float32), 1).astype(np.float64)      # bbox numbers
    b = np.round(rng.uniform(0, 1, 20000).astype(np.float32), 2).astype(np.float64)           # scores
    c = rng.standard_normal(30000) * 10.0 ** rng.integers(-30, 30, 30000)
    d = np.array([0.0, -0.0, 1.0, 1e16, 1e15, 123456789012345678.0, 1e-4, 1e-5, 0.0001234, 5e-324, 1.7976931348623157e308, 0.1, 0.5,
                  9007199254740993.0, 1e22, 1e23, 2.0 ** 60, 2.0 ** -20, float("inf"), float("-inf"), float("nan")])
    for vals in (a, b, c, d):
        assert _fmt(vals, False) == [json.dumps(float(x)) for x in vals]


def test_float32_str_is_numpys():
    rng = np.random.default_rng(1)
    f = np.concatenate([rng.uniform(-1, 1, 50000), rng.standard_normal(30000) * 10.0 ** rng.integers(-20, 20, 30000),
                        [0, 1, -1, 0.5, 1e-4, 9.9999e-5, 1e-5, 1.5e-5, 1e16, 9.99e15, 123456.0, 16777216.0, 3.4028235e38, 1e-45, 0.1, 0.3]]).astype(np.float32)
    assert _fmt(f.astype(np.float64), True) == [str(x) for x in f]


def _frame(tmp, rng, h, w, gc, sc, row_stride_extra=0, runs=40):
    W2 = w + row_stride_extra
    flow = rng.standard_normal((h, W2, 2)).astype(np.float32)
    xy = rng.uniform(-30, 1.3 * max(h, w), (2, 20, 2))
    boxes = np.concatenate([xy, xy + rng.uniform(-5, 0.9 * max(h, w), (2, 20, 2))], -1).astype(np.float32)
    boxes[0, 0] = [-0.0, 3.0, 5.04, 1e9]                     # a negative zero, a coordinate far outside
    probs = rng.uniform(0.5, 1, (2, 20)).astype(np.float32)
    probs[1, 0] = 0.005                                      # rounds to 0.0 / 0.01 boundary region
    n = gc + sc
    masks = (rng.random((max(n, 1), h, w)) > 0.5).astype(np.uint8)
    if n > 1:
        masks[1] = 0
    if n > 2:
        masks[2] = 1
    if n > 3:                                                # a first run of 44 = 12 + 32: its first character is 44 + 48 = a backslash
        m = np.zeros(h * w, np.uint8)
        m[44:50] = 1
        masks[3] = m.reshape((h, w), order="F")
    pos = [np.flatnonzero(np.diff(np.concatenate(([0], (m != 0).reshape(-1, order="F").astype(np.int8))))).astype(np.int32) for m in masks[:n]]
    off = np.concatenate(([0], np.cumsum([len(p) for p in pos]))).astype(np.int32) if n else np.zeros(1, np.int32)
    pool = np.concatenate(pos + [np.zeros(1, np.int32)]).astype(np.int32)
    conf = rng.uniform(-1, 1, max(n, 1)).astype(np.float32)
    conf[0] = 1.2e-5                                         # numpy prints this in scientific form
    return flow, boxes, probs, masks, pool, off, conf


@pytest.mark.parametrize("gc,sc,extra,flo", [(20, 20, 0, True), (3, 0, 6, True), (0, 5, 0, False), (0, 0, 0, True), (1, 1, 2, False)])
def test_a_frames_files_are_the_python_writers_bytes(tmp_path, gc, sc, extra, flo):
    lib = _lib.load()
    rng = np.random.default_rng(gc * 100 + sc)
    h, w, scale = 37, 52, 1.5609756
    flow, boxes, probs, masks, pool, off, conf = _frame(tmp_path, rng, h, w, gc, sc, extra)
    f = _lib.FrameFiles()
    out = tmp_path / "deep" / "er"
    names = ["a.flo", "g.json", "s.json", "c.json", "r.json"]
    f.flo_path = str(out / names[0]).encode() if flo else None
    f.flow, f.flow_row_stride, f.h, f.w, f.scale = flow.ctypes.data, 2 * (w + extra), h, w, float(np.float32(scale))
    for k in range(2):
        f.boxes[k], f.probs[k] = boxes[k].ctypes.data, probs[k].ctypes.data
    f.count[0], f.count[1] = gc, sc
    for k in range(4):
        f.json_path[k] = str(out / names[k + 1]).encode()
    f.conf, f.rle_pool, f.rle_offsets = conf.ctypes.data, pool.ctypes.data, off.ctypes.data
    assert lib.premvos_write_frame_files_host(C.byref(f)) == 1          # the directory does not exist: nothing is created for the caller
    os.makedirs(out)
    assert lib.premvos_write_frame_files_host(C.byref(f)) == 0
    g = results_json(boxes[0:1], probs[0:1], np.array([gc]), scale, (h, w))[0]
    s = results_json(boxes[1:2], probs[1:2], np.array([sc]), scale, (h, w))[0]
    both = g + s
    ref = [dict(p) for p in both]
    for j, q in enumerate(ref):
        q["segmentation"] = rle.encode(masks[j])
        q["conf_score"] = str(conf[j])
    expect = {"g.json": json.dumps(g), "s.json": json.dumps(s), "c.json": json.dumps(both), "r.json": json.dumps(ref)}
    for name, text in expect.items():
        assert (out / name).read_text() == text, name
    if gc + sc > 3:
        assert "\\\\" in expect["r.json"]                    # the count strings of ragged masks do contain the escaped character
    assert (out / "a.flo").exists() == flo
    if flo:
        assert (out / "a.flo").read_bytes() == flo_bytes(flow[:, :w])


def test_bad_arguments_are_errors_not_crashes(tmp_path):
    lib = _lib.load()
    f = _lib.FrameFiles()
    assert lib.premvos_write_frame_files_host(C.byref(f)) < 0            # h = w = 0
    f.h, f.w = 4, 4
    f.count[0] = 1                                                        # detections without arrays
    assert lib.premvos_write_frame_files_host(C.byref(f)) < 0 and b"detections" in lib.premvos_last_error()
    f.count[0] = 0
    f.flo_path = str(tmp_path / "x.flo").encode()                          # a flow file without flow
    assert lib.premvos_write_frame_files_host(C.byref(f)) < 0
    f.flo_path = None
    assert lib.premvos_write_frame_files_host(C.byref(f)) == 0             # nothing to write is fine


def test_box_arithmetic_matches_numpy_on_hostile_values(tmp_path):
    """eval.py:93-94 + train.py:388-428 on values a real net can emit and a synthetic one will: huge, tiny, negative, exactly on the
    clip limits, half-way cases of the rounding, infinities and NaNs -- the C twin's float32 arithmetic and number formatting against
    numpy + json (NaN / Infinity are what json.dumps writes for them, and what the reference would have written)."""
    lib = _lib.load()
    rng = np.random.default_rng(11)
    h, w, scale = 480, 854, (749 / 480 + 1333 / 854) / 2
    pool, off, conf = np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.float32)
    specials = np.array([0.0, -0.0, 0.05, 0.15, 0.25, 0.35, 1e-7, -1e-7, 1e9, -1e9, 3.4e38, -3.4e38, np.inf, -np.inf, np.nan, 854.0 * scale,
                         480.0 * scale, 854.04 * scale, 0.049999, 1234.5678, 16777216.0], np.float32)
    for rep in range(40):
        boxes = np.zeros((2, 20, 4), np.float32)
        boxes[0] = (rng.standard_normal((20, 4)) * 10.0 ** rng.integers(-3, 6, (20, 4))).astype(np.float32)
        boxes[1] = rng.choice(specials, (20, 4))
        probs = np.concatenate([rng.uniform(0, 1, (1, 20)), rng.choice([0.005, 0.015, 0.125, 0.995, 1.0, 0.0, 0.5, np.nan], (1, 20))]).astype(np.float32)
        f = _lib.FrameFiles()
        f.h, f.w, f.scale = h, w, float(np.float32(scale))
        for k in range(2):
            f.boxes[k], f.probs[k], f.count[k] = boxes[k].ctypes.data, probs[k].ctypes.data, 20
        f.json_path[0], f.json_path[1] = str(tmp_path / "g.json").encode(), str(tmp_path / "s.json").encode()
        f.conf, f.rle_pool, f.rle_offsets = conf.ctypes.data, pool.ctypes.data, off.ctypes.data
        assert lib.premvos_write_frame_files_host(C.byref(f)) == 0, lib.premvos_last_error()
        with np.errstate(all="ignore"):
            g = results_json(boxes[0:1], probs[0:1], np.array([20]), scale, (h, w))[0]
            s = results_json(boxes[1:2], probs[1:2], np.array([20]), scale, (h, w))[0]
        assert (tmp_path / "g.json").read_text() == json.dumps(g), rep
        assert (tmp_path / "s.json").read_text() == json.dumps(s), rep
