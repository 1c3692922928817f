"""configs[0] of BASELINE.json: the 3-frame 'bear' clip through the stage drivers and file formats end to end
(SURVEY 8d 'Config 1 (plumbing)'): .flo / proposal JSON / combined JSON / refined JSON exactly where and how the
unchanged MergeTrack stage (and the ReID stage, when the reference's own is used) read them, with stage-level resume."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import proposal_oracle as PO  # noqa: E402
from oracle import pwc_oracle as O  # noqa: E402
from oracle import refinement_oracle as RO  # noqa: E402
from oracle import reid_oracle as QO  # noqa: E402

BLOCKS, MIDDLE = (1, 1, 2, 1), 1
REID_UNITS = [QO.UNITS[0], ("res3", 2, (64, 64), (3, 3), (2, 1)), ("res15", 3, (32, 64, 96), (1, 3, 1), (1, 2, 1))]
REID_NETWORK = {"conv0": {"class": "Conv", "n_features": 64, "activation": "linear"},
                "res0": {"class": "ResidualUnit2", "n_features": 128, "strides": [[2, 2], [1, 1]], "from": ["conv0"]},
                "res3": {"class": "ResidualUnit2", "n_features": 64, "strides": [[2, 2], [1, 1]], "from": ["res0"]},
                "res15": {"class": "ResidualUnit2", "n_convs": 3, "n_features": [32, 64, 96],
                          "filter_size": [[1, 1], [3, 3], [1, 1]], "strides": [[1, 1], [2, 2], [1, 1]], "from": ["res3"]},
                "conv1": {"class": "Conv", "n_features": 500, "batch_norm": True, "filter_size": [3, 3], "pool_size": [3, 3],
                          "from": ["res15"]},
                "fc1": {"class": "FullyConnected", "n_features": 500, "batch_norm": True, "from": ["conv1"]},
                "fc2": {"class": "FullyConnected", "n_features": 500, "batch_norm": True, "from": ["fc1"]},
                "outputTriplet": {"class": "FullyConnectedWithTripletLoss", "n_features": 128, "batch_norm": True,
                                  "activation": "linear", "from": ["fc2"]}}


def _harness():
    """tools/run_stages.py: the dev harness that chains the stage drivers like simple_run.sh:21-68 (not product code)."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "run_stages.py")
    spec = importlib.util.spec_from_file_location("run_stages", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _make_tree(root, h=120, w=200, t=3, videos=None):
    """``videos``: {name: frames}; default one clip 'bear' of ``t`` frames (its decoded frames are returned)."""
    from PIL import Image
    videos = videos or {"bear": t}
    frames = []
    for vi, (name, nt) in enumerate(videos.items()):
        seq_dir = root / "data" / "DAVIS" / "JPEGImages" / "480p" / name
        seq_dir.mkdir(parents=True)
        for i in range(nt):
            pair = O.synth_frame_pair(h, w + (-w) % 8, seed=40 + vi, shift=(1.5 * i, -0.5 * i))
            img = (pair[0, 3:, :, :w].permute(1, 2, 0) * 255).round().to(torch.uint8).numpy()
            Image.fromarray(img).save(seq_dir / f"{i:05d}.jpg", quality=95)
            if vi == 0:
                frames.append(np.asarray(Image.open(seq_dir / f"{i:05d}.jpg").convert("RGB")))
    (root / "seq_to_run.txt").write_text("".join(f"data/DAVIS/JPEGImages/480p/{name}/\n" for name in videos))
    wd = root / "weights"
    wd.mkdir()
    torch.save({"state_dict": O.synth_state_dict(0)}, wd / "pwc.pth.tar")
    from premvos_amd import weights as W
    # general + refinement as TF tensor-bundle checkpoints (what simple_run.sh passes), specific as a torch pickle
    W.save_tf_checkpoint(str(wd / "proposal_general_weights"), W.proposal_weights_to_tf(PO.synth_weights(0, BLOCKS)))
    torch.save(PO.synth_weights(1, BLOCKS), wd / "specific.pt")
    W.save_tf_checkpoint(str(wd / "refinement_specific_weights"), W.refinement_weights_to_tf(RO.synth_weights(0, MIDDLE)))
    # ReID: config under code/ReID_net/configs/ with a relative 'load' (the reference runs that stage from code/)
    W.save_tf_checkpoint(str(wd / "ReID_general_weights"), W.reid_weights_to_tf(QO.synth_weights(0, REID_UNITS)))
    cdir = root / "code" / "ReID_net" / "configs"
    cdir.mkdir(parents=True)
    (cdir / "run").write_text("# reduced ReID net for the plumbing test\n" + json.dumps(
        {"model": "Re-ID", "load": "../weights/ReID_general_weights", "input_size": [128, 128], "network": REID_NETWORK}))
    return frames


def test_three_frame_clip_end_to_end(tmp_path):
    from premvos_amd import rle
    run_pipeline = _harness()
    from premvos_amd.flow.driver import readFlowFile
    frames = _make_tree(tmp_path)
    cwd = os.getcwd()
    args = ["--root", str(tmp_path), "--flow_weights", "weights/pwc.pth.tar", "--general_weights",
            "weights/proposal_general_weights", "--specific_weights", "weights/specific.pt", "--refinement_weights",
            "weights/refinement_specific_weights"]
    try:
        assert run_pipeline.main(args) == 0
        inter = tmp_path / "output" / "intermediate"
        # flow: named by the first frame of the pair, none for the last frame (script_pwc_multi.py:94-102)
        flos = sorted(os.listdir(inter / "flow" / "bear"))
        assert flos == ["00000.flo", "00001.flo"]
        flo = readFlowFile(str(inter / "flow" / "bear" / "00000.flo"))
        assert flo.shape == (120, 200, 2) and np.isfinite(flo).all()
        # the .flo equals the oracle chain on the decoded JPEGs (script_pwc_multi.py:33-70)
        from oracle import cv_resize_oracle as CR
        x0, h_, w_ = CR.flow_preprocess(frames[0], frames[1])
        with torch.no_grad():
            ref = CR.flow_postprocess(O.pwc_forward(O.synth_state_dict(0), torch.from_numpy(x0))[0].numpy(), 120, 200, h_, w_)
        assert np.abs(flo - ref).max() < 2e-2 * max(1.0, np.abs(ref).max())
        # proposals: one JSON per frame, bbox xywh 1 decimal, score 2 decimals; combined = general + specific
        for t in range(3):
            g = json.load(open(inter / "general_proposals" / "bear" / f"{t:05d}.json"))
            s = json.load(open(inter / "specific_proposals" / "bear" / f"{t:05d}.json"))
            c = json.load(open(inter / "combined_proposals" / "bear" / f"{t:05d}.json"))
            r = json.load(open(inter / "refined_proposals" / "bear" / f"{t:05d}.json"))
            assert c == g + s and len(r) == len(c) and len(g) <= 20 and len(s) <= 20
            for p in c:
                assert set(p) == {"bbox", "score"} and len(p["bbox"]) == 4
                assert all(round(v, 1) == v for v in p["bbox"]) and round(p["score"], 2) == p["score"]
            for p, q in zip(r, c):
                assert p["bbox"] == q["bbox"] and p["score"] == q["score"]
                assert p["segmentation"]["size"] == [120, 200] and isinstance(p["segmentation"]["counts"], str)
                assert isinstance(p["conf_score"], str) and -1.0 <= float(p["conf_score"]) <= 1.0
                m = rle.decode(p["segmentation"])
                assert m.shape == (120, 200) and set(np.unique(m)) <= {0, 1}
        # refined result of frame 0 equals the oracle forwarder on the same combined JSON
        c0 = json.load(open(inter / "combined_proposals" / "bear" / "00000.json"))
        r0 = json.load(open(inter / "refined_proposals" / "bear" / "00000.json"))
        ref0 = RO.refine_proposals(RO.synth_weights(0, MIDDLE), frames[0], c0[:3], MIDDLE)
        for a, b in zip(r0[:3], ref0):
            assert abs(float(a["conf_score"]) - float(b["conf_score"])) < 2e-3
            assert (rle.decode(a["segmentation"]) != RO.rle_decode(b["segmentation"])).mean() < 5e-3
        # ReID stage: every refined proposal with a non-empty mask gains a 128-d embedding (ReIDForwarding.py:68-74)
        for t in range(3):
            r = json.load(open(inter / "refined_proposals" / "bear" / f"{t:05d}.json"))
            q = json.load(open(inter / "ReID_proposals" / "bear" / f"{t:05d}.json"))
            assert len(q) == len(r)
            for a, b in zip(q, r):
                assert {k: v for k, v in a.items() if k != "ReID"} == b
                bb = rle.to_bbox(b["segmentation"])
                assert ("ReID" in a) == (bb[2] > 0 and bb[3] > 0)
                assert "ReID" not in a or (len(a["ReID"]) == 128 and np.isfinite(a["ReID"]).all())
        q0 = json.load(open(inter / "ReID_proposals" / "bear" / "00000.json"))
        have = [p for p in q0 if "ReID" in p][:2]
        if have:
            cb = QO.context_boxes([rle.to_bbox(p["segmentation"]) for p in have], 120, 200, feed=False)
            ref = QO.forward(QO.synth_weights(0, REID_UNITS), np.stack([QO.make_crop(frames[0], b, feed=False) for b in cb]),
                             REID_UNITS)
            for p, e in zip(have, ref):
                assert np.abs(np.array(p["ReID"]) - e).max() < 1e-3 * max(1.0, np.abs(ref).max())
        assert not (tmp_path / "output" / "final").exists()      # MergeTrack stays the reference's (out of scope)
        # stage-level resume: nothing is recomputed when the directories exist (simple_run.sh:23,30,38,46,53)
        before = {p: os.path.getmtime(p) for p in map(str, inter.rglob("*")) if os.path.isfile(p)}
        assert run_pipeline.main(args) == 0
        assert before == {p: os.path.getmtime(p) for p in before}
    finally:
        os.chdir(cwd)


def test_batched_drivers_equal_frame_by_frame_drivers(tmp_path, monkeypatch):
    """PREMVOS_DRIVER_BATCH only changes how many independent frames share a launch list: same .flo payloads, same proposal
    and refined JSON up to fp32 round-off (another batch size may pick another k-split)."""
    from premvos_amd import rle
    run_pipeline = _harness()
    from premvos_amd.flow.driver import readFlowFile
    roots = []
    cwd = os.getcwd()
    try:
        for tag, batch in (("one", "1"), ("many", "8")):
            root = tmp_path / tag
            root.mkdir()
            _make_tree(root)
            monkeypatch.setenv("PREMVOS_DRIVER_BATCH", batch)
            assert run_pipeline.main(["--root", str(root), "--flow_weights", "weights/pwc.pth.tar", "--general_weights",
                                      "weights/proposal_general_weights", "--specific_weights", "weights/specific.pt",
                                      "--refinement_weights", "weights/refinement_specific_weights"]) == 0
            os.chdir(cwd)
            roots.append(root / "output" / "intermediate")
    finally:
        os.chdir(cwd)
    a, b = roots
    for t in range(2):
        fa, fb = (readFlowFile(str(r / "flow" / "bear" / f"{t:05d}.flo")) for r in (a, b))
        assert np.abs(fa - fb).max() < 1e-4
    for t in range(3):
        for stage in ("general_proposals", "specific_proposals"):
            pa, pb = (json.load(open(r / stage / "bear" / f"{t:05d}.json")) for r in (a, b))
            assert len(pa) == len(pb)
            for x, y in zip(pa, pb):
                assert np.abs(np.array(x["bbox"]) - np.array(y["bbox"])).max() <= 0.1001 and abs(x["score"] - y["score"]) <= 0.0101
        ra, rb = (json.load(open(r / "refined_proposals" / "bear" / f"{t:05d}.json")) for r in (a, b))
        assert len(ra) == len(rb)
        for x, y in zip(ra, rb):
            if x["bbox"] == y["bbox"]:
                assert abs(float(x["conf_score"]) - float(y["conf_score"])) < 1e-4
                assert (rle.decode(x["segmentation"]) != rle.decode(y["segmentation"])).mean() < 1e-3


def test_overlapped_drivers_write_the_same_bytes_as_serial_drivers(tmp_path, monkeypatch):
    """premvos_amd.io_pipeline (decode-ahead thread pool, two GPU lanes, background writer) only reorders host work: every
    .flo / proposal JSON / refined JSON / ReID JSON is byte-identical to the serial drivers' (PREMVOS_IO_THREADS=0, LANES=1)."""
    run_stages = _harness()
    roots = []
    cwd = os.getcwd()
    try:
        for tag, threads, nl in (("serial", "0", "1"), ("overlap", "3", "2")):
            root = tmp_path / tag
            root.mkdir()
            _make_tree(root, t=5)
            monkeypatch.setenv("PREMVOS_IO_THREADS", threads)
            monkeypatch.setenv("PREMVOS_IO_LANES", nl)
            monkeypatch.setenv("PREMVOS_DRIVER_BATCH", "2")
            assert run_stages.main(["--root", str(root), "--flow_weights", "weights/pwc.pth.tar", "--general_weights",
                                    "weights/proposal_general_weights", "--specific_weights", "weights/specific.pt",
                                    "--refinement_weights", "weights/refinement_specific_weights"]) == 0
            os.chdir(cwd)
            roots.append(root / "output" / "intermediate")
    finally:
        os.chdir(cwd)
    a, b = roots
    files = sorted(str(p.relative_to(a)) for p in a.rglob("*") if p.is_file())
    assert files == sorted(str(p.relative_to(b)) for p in b.rglob("*") if p.is_file())
    assert any(f.endswith(".flo") for f in files) and sum(f.startswith("refined_proposals") for f in files) == 5
    for f in files:
        assert (a / f).read_bytes() == (b / f).read_bytes(), f


def test_streaming_driver_writes_the_stage_drivers_tree(tmp_path, monkeypatch):
    """premvos_amd.stream (one decode per frame, flow / proposals x2 / combine / refinement on three host threads) against the
    four stage drivers run one after the other like simple_run.sh: same files, same bytes (at matched launch batch sizes)."""
    from premvos_amd import stream
    run_stages = _harness()
    cwd = os.getcwd()
    monkeypatch.setenv("PREMVOS_DRIVER_BATCH", "2")
    a_root, b_root = tmp_path / "stages", tmp_path / "stream"
    args = ["--flow_weights", "weights/pwc.pth.tar", "--general_weights", "weights/proposal_general_weights",
            "--specific_weights", "weights/specific.pt", "--refinement_weights", "weights/refinement_specific_weights"]
    try:
        for root in (a_root, b_root):
            root.mkdir()
            _make_tree(root, t=5)
        assert run_stages.main(["--root", str(a_root)] + args) == 0
        os.chdir(cwd)
        assert stream.main(["--root", str(b_root), "--batch", "2"] + args) == 0
    finally:
        os.chdir(cwd)
    a, b = a_root / "output" / "intermediate", b_root / "output" / "intermediate"
    stages = ("flow", "general_proposals", "specific_proposals", "combined_proposals", "refined_proposals")
    fa = sorted(str(p.relative_to(a)) for p in a.rglob("*") if p.is_file() and str(p.relative_to(a)).split("/")[0] in stages)
    fb = sorted(str(p.relative_to(b)) for p in b.rglob("*") if p.is_file())
    assert fa == fb and len(fa) == 4 + 4 * 5
    for f in fa:
        assert (a / f).read_bytes() == (b / f).read_bytes(), f


def _stream_subprocess(root, *extra, gpus=1):
    """A FRESH process (or two ranks of them, sharing the test box's one GPU over gloo -- RCCL needs a device per rank)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"PREMVOS_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PYTHONPATH": repo})
    r = subprocess.run([sys.executable, "-m", "premvos_amd.stream", "--root", str(root), "--gpus", str(gpus), "--flow_weights",
                        "weights/pwc.pth.tar", "--general_weights", "weights/proposal_general_weights", "--specific_weights",
                        "weights/specific.pt", "--refinement_weights", "weights/refinement_specific_weights", *extra],
                       capture_output=True, text=True, env=env, timeout=1500, cwd=repo)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


def _same_tree(a, b, n_files):
    fa = sorted(str(p.relative_to(a)) for p in a.rglob("*") if p.is_file())
    fb = sorted(str(p.relative_to(b)) for p in b.rglob("*") if p.is_file())
    assert fa == fb and len(fa) == n_files, (len(fa), len(fb))
    for f in fa:
        assert (a / f).read_bytes() == (b / f).read_bytes(), f


def test_two_ranks_write_the_one_rank_tree_whole_videos(tmp_path):
    """`python -m premvos_amd.stream --gpus 2` on a 7-frame, 2-video tree (one whole video per rank,
    premvos_amd.parallel.plan_shards) writes the byte-identical output tree of the 1-rank run -- both in fresh processes, so this
    is also the run-to-run reproducibility check (the conv configurations come from the shipped table or the
    closed-form rule, never from a stopwatch race between kernels that round differently)."""
    roots = []
    for tag, gpus in (("one", 1), ("two", 2)):
        root = tmp_path / tag
        root.mkdir()
        _make_tree(root, videos={"bear": 4, "camel": 3})
        out = _stream_subprocess(root, "--batch", "2", gpus=gpus)
        assert "frames: 7" in out
        roots.append(root / "output")
    _same_tree(roots[0] / "intermediate", roots[1] / "intermediate", (3 + 2) + 4 * 7)
    m1, m2 = (json.load(open(r / "premvos_amd_manifest.json")) for r in roots)
    assert m1["ranks"] == 1 and m2["ranks"] == 2 and m2["frames"] == 7
    assert sorted(len(p) for p in m2["shards"]) == [1, 1]                                   # one whole video per rank
    c1, c2 = m1["conv_configurations"], m2["conv_configurations"]
    assert c1["table_sha256_16"] == c2["table_sha256_16"] and c1["table_sha256_16"]        # the same shipped table decided
    assert c1["signatures_explored_by_time"] == 0 and c2["signatures_explored_by_time"] == 0


def test_two_ranks_write_the_one_rank_tree_frame_ranges(tmp_path):
    """Fewer videos than ranks: ONE 7-frame video in chunks of 2 -> rank 0 owns frames [0,4) and reads frame 4 as the second
    image of its last flow pair, rank 1 owns [4,7).  Same bytes as the 1-rank run, with per-rank writers, with --gather (every
    rank's results handed to rank 0 in one gather per round of chunks) and with a merge rank that computes nothing itself
    (--merge-share 0: rank 0 only takes part in the gathers and writes rank 1's files)."""
    roots = []
    for tag, gpus, extra in (("one", 1, ()), ("two", 2, ()), ("gathered", 2, ("--gather",)),
                             ("merge_only", 2, ("--gather", "--merge-share", "0"))):
        root = tmp_path / tag
        root.mkdir()
        _make_tree(root, t=7)
        _stream_subprocess(root, "--batch", "2", *extra, gpus=gpus)
        roots.append(root / "output")
    for other in roots[1:]:
        _same_tree(roots[0] / "intermediate", other / "intermediate", 6 + 4 * 7)
    m = json.load(open(roots[1] / "premvos_amd_manifest.json"))
    assert sorted((a, b) for p in m["shards"] for _, a, b in p) == [(0, 4), (4, 7)]
    m = json.load(open(roots[3] / "premvos_amd_manifest.json"))
    assert m["merge_share"] == 0.0 and m["shards"][0] == [] and [(a, b) for _, a, b in m["shards"][1]] == [(0, 7)]


def test_binary_side_car_holds_what_the_json_holds(tmp_path):
    """PREMVOS_SIDECAR fast path: the refinement stage writes <frame>.pmv (bit-packed masks from the GPU, no RLE
    strings), the ReID stage reads and extends it; converted back (premvos_amd.sidecar) every file is the JSON the default path
    writes -- same floats, same RLE strings, same conf_score strings; ReID vectors agree to fp32 round-off of the batch."""
    from PIL import Image
    from premvos_amd import sidecar as sc
    from premvos_amd.refinement import driver as rd
    from premvos_amd.reid import driver as qd
    from premvos_amd.reid import ReIDEngine, ReIDNet
    h, w = 120, 200
    rng = np.random.default_rng(3)
    (tmp_path / "img" / "seq").mkdir(parents=True)
    (tmp_path / "bb" / "seq").mkdir(parents=True)
    for t in range(3):
        pair = O.synth_frame_pair(h, w, seed=41, shift=(2.0 * t, 1.0 * t))
        img = (pair[0, 3:, :, :w].permute(1, 2, 0) * 255).round().to(torch.uint8).numpy()
        Image.fromarray(img).save(tmp_path / "img" / "seq" / f"{t:05d}.jpg", quality=95)
        n = (5, 0, 3)[t]
        props = [{"bbox": [round(float(x), 1) for x in (rng.uniform(0, 90), rng.uniform(0, 40), rng.uniform(30, 100), rng.uniform(30, 70))],
                  "score": round(float(rng.uniform(0.5, 1)), 2)} for _ in range(n)]
        (tmp_path / "bb" / "seq" / f"{t:05d}.json").write_text(json.dumps(props))
    engine = rd.RefinementEngine(rd.RefinementNet(RO.synth_weights(1, MIDDLE), MIDDLE))
    args = (str(tmp_path / "img") + "/", str(tmp_path / "bb") + "/")
    assert rd.forward_directory(engine, *args, str(tmp_path / "ref_json") + "/", sidecar=False) == 3
    assert rd.forward_directory(engine, *args, str(tmp_path / "ref_pmv") + "/", sidecar=True) == 3
    assert sorted(os.listdir(tmp_path / "ref_pmv" / "seq")) == [f"{t:05d}{sc.EXT}" for t in range(3)]
    assert sc.convert_tree(str(tmp_path / "ref_pmv"), str(tmp_path / "ref_back")) == 3
    for t in range(3):
        a = (tmp_path / "ref_json" / "seq" / f"{t:05d}.json").read_text()
        b = (tmp_path / "ref_back" / "seq" / f"{t:05d}.json").read_text()
        assert a == b, t                                                   # byte for byte
    assert os.path.getsize(tmp_path / "ref_pmv" / "seq" / "00000.pmv") == 24 + 5 * (32 + 8 + 4 + h * w // 8)
    # ReID stage on either format
    q = ReIDEngine(ReIDNet(QO.synth_weights(0, REID_UNITS), units=[(n_, f, k, s) for n_, _, f, k, s in REID_UNITS]))
    assert qd.forward_directory(q, args[0], str(tmp_path / "ref_json") + "/", str(tmp_path / "reid_json") + "/") == 3
    assert qd.forward_directory(q, args[0], str(tmp_path / "ref_pmv") + "/", str(tmp_path / "reid_pmv") + "/") == 3
    sc.convert_tree(str(tmp_path / "reid_pmv"), str(tmp_path / "reid_back"))
    for t in range(3):
        a = json.load(open(tmp_path / "reid_json" / "seq" / f"{t:05d}.json"))
        b = json.load(open(tmp_path / "reid_back" / "seq" / f"{t:05d}.json"))
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert {k: v for k, v in x.items() if k != "ReID"} == {k: v for k, v in y.items() if k != "ReID"}
            assert ("ReID" in x) == ("ReID" in y)
            if "ReID" in x:
                assert np.abs(np.array(x["ReID"]) - np.array(y["ReID"])).max() < 1e-5 * max(1.0, np.abs(x["ReID"]).max())


def test_acceptance_command_runs_end_to_end_on_a_synthetic_tree(tmp_path):
    """tools/accept_davis.py on a tree laid out like the released one -- weights/PReMVOS_weights/... as TF
    tensor-bundle checkpoints + the PWC pickle, data/DAVIS/{JPEGImages,Annotations}/480p -- with synthetic weights and frames:
    step 1 reads every checkpoint, step 2 runs the streaming driver in a fresh process, step 3 (MergeTrack, the reference's own
    program) is stood in for by an output/final/ that equals the annotations, step 4 evaluates J / F against them."""
    import importlib.util
    import sys
    from PIL import Image
    from premvos_amd import weights as W
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("accept_davis", os.path.join(repo, "tools", "accept_davis.py"))
    A = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(A)
    root = tmp_path / "premvos"
    root.mkdir()
    _make_tree(root, t=4)
    wp = root / "weights" / "PReMVOS_weights"
    for sub in ("optical_flow_net", "proposal_net/general_weights", "proposal_net/specific_weights", "refinement_net/specific_weights"):
        (wp / sub).mkdir(parents=True)
    os.replace(root / "weights" / "pwc.pth.tar", wp / "optical_flow_net" / "pwc_net.pth.tar")
    W.save_tf_checkpoint(str(wp / "proposal_net/general_weights/proposal_general_weights"), W.proposal_weights_to_tf(PO.synth_weights(0, BLOCKS)))
    W.save_tf_checkpoint(str(wp / "proposal_net/specific_weights/proposal_specific_weights"), W.proposal_weights_to_tf(PO.synth_weights(1, BLOCKS)),
                         num_shards=2)
    W.save_tf_checkpoint(str(wp / "refinement_net/specific_weights/refinement_specific_weights"),
                         W.refinement_weights_to_tf(RO.synth_weights(0, MIDDLE)))
    pal = [0, 0, 0, 128, 0, 0] + [0] * (3 * 254)
    for sub in ("data/DAVIS/Annotations/480p/bear", "output/final/bear"):
        (root / sub).mkdir(parents=True)
        for t in range(4):
            g = np.zeros((120, 200), np.uint8)
            g[30 + 2 * t:80 + 2 * t, 50 + 3 * t:120 + 3 * t] = 1
            im = Image.frombytes("P", (200, 120), g.tobytes())
            im.putpalette(pal)
            im.save(root / sub / f"{t:05d}.png")
    cwd = os.getcwd()
    try:
        rc = A.main(["--root", str(root), "--skip-reid", "--tolerance", "100"])
    finally:
        os.chdir(cwd)
    assert rc == 0
    inter = root / "output" / "intermediate"
    assert len(list((inter / "flow" / "bear").glob("*.flo"))) == 3
    assert len(list((inter / "refined_proposals" / "bear").glob("*.json"))) == 4
    r = json.loads((root / "output" / "premvos_amd_davis_eval.json").read_text())
    assert r["mean_J"] == 1.0 and r["mean_F"] == 1.0 and r["sequences"] == 1
    assert json.loads((root / "output" / "premvos_amd_manifest.json").read_text())["frames"] == 4


def test_eight_ranks_write_the_one_rank_tree_both_sharding_branches(tmp_path):
    """World 8 = one MI355X node, the ranks sharing the test box's one GPU over gloo: (a) a ragged
    3-video tree (5 / 3 / 2 frames, chunks of 2) -- fewer videos than ranks, so every video is cut into chunk-aligned frame ranges,
    most ranks own one chunk, some own none, the boundary frames are read twice; (b) the same with --gather (ranks without a
    shard item still take part in every flush); (c) a 9-video tree -- at least as many videos as ranks: whole videos, one rank
    gets two.  Every tree is byte-identical to the 1-rank run's."""
    trees = {"ragged": {"bear": 5, "camel": 3, "dog": 2}, "many": {f"v{i}": 1 + (i % 2) for i in range(9)}}
    for name, videos in trees.items():
        n_frames = sum(videos.values())
        n_files = sum(v - 1 for v in videos.values()) + 4 * n_frames
        roots = []
        for tag, gpus, extra in [("one", 1, ()), ("eight", 8, ())] + ([("gathered", 8, ("--gather",))] if name == "ragged" else []):
            root = tmp_path / f"{name}_{tag}"
            root.mkdir()
            _make_tree(root, videos=videos)
            out = _stream_subprocess(root, "--batch", "2", *extra, gpus=gpus)
            assert f"frames: {n_frames}" in out
            roots.append(root / "output")
        for other in roots[1:]:
            _same_tree(roots[0] / "intermediate", other / "intermediate", n_files)
        m = json.load(open(roots[1] / "premvos_amd_manifest.json"))
        assert m["ranks"] == 8 and m["frames"] == n_frames and len(m["shards"]) == 8
        if name == "ragged":
            assert sum(1 for p in m["shards"] if not p) >= 1            # 3 + 2 + 1 = 6 chunks... some rank owns nothing
            assert sorted((v.rstrip("/").split("/")[-1], a, b) for p in m["shards"] for v, a, b in p) == \
                [("bear", 0, 2), ("bear", 2, 4), ("bear", 4, 5), ("camel", 0, 2), ("camel", 2, 3), ("dog", 0, 2)]
        else:
            assert sorted(len(p) for p in m["shards"]) == [1] * 7 + [2]


def test_streaming_driver_in_the_bf16x3_mode_writes_the_fp32_trees_proposals_and_masks(tmp_path, monkeypatch):
    """PREMVOS_PRECISION=bf16x3 through the file-to-file product path (fresh processes): the nets run on the resident S8 layout
    (packed refinement slots, eager launches, two lanes), the output tree has the fp32 run's files; proposal JSONs are equal or differ
    in the last printed digit of a box / score, and the refined masks agree with the fp32 run's to IoU >= 0.98 (fp32-class arithmetic:
    a mask pixel flips only where the posterior is within ~1e-4 of 0.5)."""
    import subprocess
    import sys
    from premvos_amd import rle
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    roots = {}
    for prec in ("fp32", "bf16x3"):
        root = tmp_path / prec
        root.mkdir()
        _make_tree(root, t=5)
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
        env.update({"PREMVOS_PRECISION": prec, "PYTHONPATH": repo})
        r = subprocess.run([sys.executable, "-m", "premvos_amd.stream", "--root", str(root), "--batch", "2", "--flow_weights",
                            "weights/pwc.pth.tar", "--general_weights", "weights/proposal_general_weights", "--specific_weights",
                            "weights/specific.pt", "--refinement_weights", "weights/refinement_specific_weights"],
                           capture_output=True, text=True, env=env, timeout=900, cwd=repo)
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
        roots[prec] = root / "output" / "intermediate"
    fa = sorted(str(p.relative_to(roots["fp32"])) for p in roots["fp32"].rglob("*") if p.is_file())
    fb = sorted(str(p.relative_to(roots["bf16x3"])) for p in roots["bf16x3"].rglob("*") if p.is_file())
    assert fa == fb and len(fa) == 4 + 4 * 5
    n_masks = 0
    for f in fa:
        if not f.startswith("refined_proposals"):
            continue
        a, b = json.load(open(roots["fp32"] / f)), json.load(open(roots["bf16x3"] / f))
        assert len(a) == len(b)
        for p, q in zip(a, b):
            assert np.abs(np.array(p["bbox"]) - np.array(q["bbox"])).max() <= 0.11 and abs(p["score"] - q["score"]) <= 0.011
            ma, mb = rle.decode(p["segmentation"]), rle.decode(q["segmentation"])
            inter, union = int((ma & mb).sum()), int((ma | mb).sum())
            assert union == 0 or inter / union >= 0.98, (f, inter, union)
            assert abs(float(p["conf_score"]) - float(q["conf_score"])) < 1e-2
            n_masks += 1
    # (the reduced-depth random-weight nets of this tree may detect nothing; the flow files always exist: PWC-Net on the on-the-fly
    #  bf16x3 kernel against the fp32 kernels)
    from premvos_amd.flow.driver import readFlowFile
    n_flo = 0
    for f in fa:
        if f.endswith(".flo"):
            x, y = readFlowFile(str(roots["fp32"] / f)), readFlowFile(str(roots["bf16x3"] / f))
            assert np.abs(x - y).max() < 1e-3 * max(1.0, np.abs(x).max()), f
            n_flo += 1
    assert n_flo == 4 and n_masks >= 0
