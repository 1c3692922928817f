"""premvos_amd.sidecar: the optional binary proposal side-car holds exactly what the reference's proposal JSON
holds -- converting it back gives the same JSON text (floats, RLE strings, conf_score strings, ReID lists)."""
import json
import os

import numpy as np
import pytest

from premvos_amd import rle
from premvos_amd import sidecar as sc


def _props(rng, h, w, n):
    masks = (rng.random((n, h, w)) > 0.55).astype(np.uint8)
    if n:
        masks[n // 2] = 0                                             # an empty mask (ReID skips it)
    props = []
    for i in range(n):
        p = {"bbox": [round(float(v), 1) for v in rng.uniform(0, 40, 4)], "score": round(float(rng.uniform(0.5, 1)), 2),
             "segmentation": rle.encode(masks[i]), "conf_score": str(np.float32(rng.uniform(-1, 1)))}
        if i % 2 == 0 and i != n // 2:
            p["ReID"] = rng.standard_normal(128).astype(np.float32).tolist()
        props.append(p)
    return props, masks


@pytest.mark.parametrize("h,w,n", [(37, 53, 5), (48, 64, 1), (20, 31, 0)])
def test_side_car_round_trip_is_the_same_json(tmp_path, h, w, n):
    props, masks = _props(np.random.default_rng(h), h, w, n)
    d = sc.from_proposals(props, h, w)
    fn = str(tmp_path / ("a" + sc.EXT))
    sc.write_dict(fn, d)
    back = sc.read(fn)
    assert json.dumps(sc.to_proposals(back)) == json.dumps(props)
    if n:
        assert np.array_equal(sc.unpack_masks(back), masks) and np.array_equal(sc.pack_masks(masks), back["mask_bits"])
        has_reid = any("ReID" in p for p in props)
        assert os.path.getsize(fn) == 24 + n * (32 + 8 + 4 + (h * w + 7) // 8 + (513 if has_reid else 0))
        assert [sc.tight_bbox(m) for m in masks] == [rle.to_bbox(p["segmentation"]) for p in props]


def test_reader_refuses_foreign_and_truncated_files(tmp_path):
    fn = tmp_path / "x.pmv"
    fn.write_bytes(b"JSON" + bytes(40))
    with pytest.raises(ValueError, match="not a proposal side-car"):
        sc.read(str(fn))
    props, _ = _props(np.random.default_rng(1), 16, 16, 2)
    good = tmp_path / "g.pmv"
    sc.write_dict(str(good), sc.from_proposals(props, 16, 16))
    raw = good.read_bytes()
    (tmp_path / "t.pmv").write_bytes(raw + b"\0")
    with pytest.raises(ValueError, match="trailing"):
        sc.read(str(tmp_path / "t.pmv"))
    (tmp_path / "v.pmv").write_bytes(raw[:4] + (9).to_bytes(4, "little") + raw[8:])
    with pytest.raises(ValueError, match="version"):
        sc.read(str(tmp_path / "v.pmv"))


def test_convert_tree_writes_the_reference_layout(tmp_path):
    props, _ = _props(np.random.default_rng(2), 24, 40, 3)
    (tmp_path / "in" / "seq").mkdir(parents=True)
    sc.write_dict(str(tmp_path / "in" / "seq" / "00007.pmv"), sc.from_proposals(props, 24, 40))
    assert sc.main(["--to-json", str(tmp_path / "in"), str(tmp_path / "out")]) == 0
    assert json.load(open(tmp_path / "out" / "seq" / "00007.json")) == props
