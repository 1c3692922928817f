"""TF tensor-bundle reader/writer + variable-name maps (no TensorFlow, no GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import proposal_oracle as PO
from oracle import refinement_oracle as RO
from premvos_amd import weights as W


def test_crc32c_known_answers():
    assert W.crc32c(b"") == 0 and W.crc32c(b"123456789") == 0xE3069283          # CRC-32C check value
    assert W.crc32c(bytes(32)) == 0x8A9136AA                                      # RFC 3720 B.4: 32 zero bytes


def test_snappy_decompress():
    # literal "abcd", copy(offset 4, len 8) (overlapping run), literal "Z"
    stream = bytes([13]) + bytes([(4 - 1) << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4]) + bytes([0 << 2]) + b"Z"
    assert W.snappy_decompress(stream) == b"abcdabcdabcdZ"
    with pytest.raises(ValueError):
        W.snappy_decompress(bytes([99]) + bytes([0]) + b"a")


def test_bundle_roundtrip_multi_block(tmp_path):
    rng = np.random.default_rng(0)
    v = {f"scope_{i:03d}/layer/W": rng.standard_normal((3, 3, 4, 5)).astype(np.float32) for i in range(150)}
    v["global_step"] = np.array(1234, np.int64)
    v["a/very/long/name/" + "x" * 300] = np.arange(7, dtype=np.int32)
    v["empty"] = np.zeros((0, 4), np.float32)
    prefix = str(tmp_path / "ckpt")
    W.save_tf_checkpoint(prefix, v, block_size=512)
    raw = open(prefix + ".index", "rb").read()
    assert raw[-8:] == (0xDB4775248B80FB57).to_bytes(8, "little") and len(raw) > 4096       # several data blocks
    back = W.load_tf_checkpoint(prefix, verify=True)
    assert set(back) == set(v)
    for k in v:
        assert back[k].dtype == v[k].dtype and back[k].shape == v[k].shape and np.array_equal(back[k], v[k])
    # corruption is detected
    bad = bytearray(raw)
    bad[10] ^= 0xFF
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        W.read_table(prefix + ".index")
    open(prefix + ".index", "wb").write(raw[:-1])
    with pytest.raises(ValueError):
        W.read_table(prefix + ".index")


def test_bundle_multi_shard_restart_points_and_damaged_files(tmp_path):
    """What real TF-written bundles contain that the single-shard writer of round 1-2 never emitted: variables
    spread over several ``.data-0000k-of-0000N`` shards, long runs of prefix-compressed keys between restart points (restart
    interval 1 ... 64), and the failure modes of a download that went wrong: a missing shard, a truncated shard, a truncated or
    bit-flipped index, an entry whose size contradicts its shape -- each an explicit error that names the variable / file."""
    rng = np.random.default_rng(1)
    v = {f"group{g}/block{b}/conv{c}/W": rng.standard_normal((1, 1, 8, 8)).astype(np.float32)
         for g in range(4) for b in range(6) for c in (1, 2, 3)}
    v.update({f"group{g}/block{b}/conv{c}/bn/mean/EMA": rng.standard_normal(8).astype(np.float32)
              for g in range(4) for b in range(6) for c in (1, 2, 3)})
    v["global_step"] = np.array(7, np.int64)
    for shards, restart in ((3, 1), (2, 64), (5, 16)):
        prefix = str(tmp_path / f"ckpt_{shards}_{restart}")
        W.save_tf_checkpoint(prefix, v, block_size=700, restart_interval=restart, num_shards=shards)
        assert sorted(f for f in os.listdir(tmp_path) if f.startswith(f"ckpt_{shards}_{restart}.data")) == \
            [f"ckpt_{shards}_{restart}.data-{k:05d}-of-{shards:05d}" for k in range(shards)]
        back = W.load_tf_checkpoint(prefix)                                  # verify=True is the default
        assert set(back) == set(v) and all(np.array_equal(back[k], v[k]) for k in v)
    prefix = str(tmp_path / "ckpt_3_1")
    shard1 = prefix + ".data-00001-of-00003"
    raw1 = open(shard1, "rb").read()
    os.rename(shard1, shard1 + ".away")
    with pytest.raises(FileNotFoundError, match="shard 1 of 3"):
        W.load_tf_checkpoint(prefix)
    open(shard1, "wb").write(raw1[:len(raw1) // 2])
    with pytest.raises(ValueError, match="beyond the end of shard 1"):
        W.load_tf_checkpoint(prefix)
    flipped = bytearray(raw1)
    flipped[5] ^= 1
    open(shard1, "wb").write(bytes(flipped))
    with pytest.raises(ValueError, match="tensor checksum mismatch"):
        W.load_tf_checkpoint(prefix)
    assert len(W.load_tf_checkpoint(prefix, verify=False)) == len(v)         # (opting out of the checksums is possible, not default)
    open(shard1, "wb").write(raw1)
    idx = open(prefix + ".index", "rb").read()
    for cut in (len(idx) // 3, len(idx) - 20, 47):
        open(prefix + ".index", "wb").write(idx[:cut] + (idx[-48:] if cut < len(idx) - 48 else b""))
        with pytest.raises(ValueError):
            W.load_tf_checkpoint(prefix)
    open(prefix + ".index", "wb").write(idx)
    assert len(W.load_tf_checkpoint(prefix)) == len(v)


def _same(a, b):
    assert set(a) == set(b)
    for k in a:
        if isinstance(a[k], dict):
            for f in a[k]:
                assert torch.equal(a[k][f], b[k][f]), (k, f)
        else:
            assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k


def test_proposal_name_and_layout_map(tmp_path):
    w = PO.synth_weights(0, (1, 1, 2, 1))
    tfv = W.proposal_weights_to_tf(w)
    # tensorpack names / TF layouts
    assert tfv["conv0/W"].shape == (7, 7, 3, 64) and tfv["group2/block1/conv2/W"].shape == (3, 3, 256, 256)
    assert tfv["rpn/class/W"].shape == (1, 1, 1024, 15) and tfv["fastrcnn/class/W"].shape == (2048, 2)
    assert tfv["maskrcnn/deconv/W"].shape == (2, 2, 256, 2048)
    assert {"conv0/bn/gamma", "conv0/bn/beta", "conv0/bn/mean/EMA", "conv0/bn/variance/EMA"} <= set(tfv)
    prefix = str(tmp_path / "proposal_general_weights")
    W.save_tf_checkpoint(prefix, tfv)
    _same(W.load_any(prefix, "proposal"), w)


def test_refinement_name_and_layout_map(tmp_path):
    w = RO.synth_weights(0, 1)
    tfv = W.refinement_weights_to_tf(w)
    assert tfv["xception_65/entry_flow/conv1_1/weights"].shape == (3, 3, 4, 32)          # 4-channel stem (Saver.py:103-106)
    k = "xception_65/middle_flow/block1/unit_1/xception_module/separable_conv2_depthwise/depthwise_weights"
    assert tfv[k].shape == (3, 3, 728, 1)
    assert "xception_65/entry_flow/conv1_1/BatchNorm/moving_variance" in tfv
    assert tfv["logits/features/weights"].shape == (1, 1, 256, 2) and tfv["logits/features/biases"].shape == (2,)
    assert tfv["decoder/decoder_conv0_depthwise/depthwise_weights"].shape == (3, 3, 304, 1)
    prefix = str(tmp_path / "refinement_specific_weights")
    W.save_tf_checkpoint(prefix, tfv)
    _same(W.load_any(prefix, "refinement"), w)
    torch.save(w, str(tmp_path / "w.pt"))
    _same(W.load_any(str(tmp_path / "w.pt"), "refinement"), w)


def test_load_any_errors_are_explicit(tmp_path):
    """a missing file, a checkpoint with missing variables and a corrupted tensor are reported as such."""
    import pytest
    import torch
    from premvos_amd import synth
    from premvos_amd import weights as W
    with pytest.raises(FileNotFoundError, match="neither the TF bundle"):
        W.load_any(str(tmp_path / "nope"), "proposal")
    w = synth.proposal_weights(0, (1, 1, 1, 1))
    tfv = W.proposal_weights_to_tf(w)
    # optimizer slots and bookkeeping variables of a training checkpoint are ignored
    tfv["conv0/W/Momentum"] = np.zeros_like(tfv["conv0/W"])
    tfv["global_step"] = np.array(7, np.int64)
    W.save_tf_checkpoint(str(tmp_path / "full"), tfv)
    back = W.load_any(str(tmp_path / "full"), "proposal")
    assert sorted(k for k in back if not k.startswith("maskrcnn/")) == sorted(k for k in w if not k.startswith("maskrcnn/"))
    del tfv["group2/block0/conv2/W"], tfv["rpn/box/b"]
    W.save_tf_checkpoint(str(tmp_path / "partial"), tfv)
    with pytest.raises(KeyError, match="group2/block0/conv2/W"):
        W.load_any(str(tmp_path / "partial"), "proposal")
    # a flipped byte in the data shard fails the (now default) tensor checksum
    shard = [p for p in os.listdir(tmp_path) if p.startswith("full.data")][0]
    raw = bytearray(open(tmp_path / shard, "rb").read())
    raw[100] ^= 0xFF
    open(tmp_path / shard, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        W.load_any(str(tmp_path / "full"), "proposal")
    torch.save([1, 2], tmp_path / "list.pt")
    with pytest.raises(ValueError, match="pickled dict"):
        W.load_any(str(tmp_path / "list.pt"), "refinement")


def test_importers_ignore_optimizer_slots_and_bookkeeping_variables():
    """Real TF checkpoints carry more than the inference variables (global_step, learning_rate, Momentum / Adam slots named
    <variable>/<slot>): the name maps take what the nets need and drop the rest."""
    import numpy as np
    from premvos_amd import synth
    from premvos_amd import weights as W
    w = synth.proposal_weights(0, (1, 1, 1, 1))
    tfv = W.proposal_weights_to_tf(w)
    want = sorted(W.proposal_weights_from_tf(tfv))
    tfv.update({"global_step": np.array(5, np.int64), "learning_rate": np.array(0.003, np.float32),
                "conv0/W/Momentum": np.zeros_like(tfv["conv0/W"]), "group0/block0/conv1/bn/gamma/Momentum": np.zeros(64, np.float32)})
    assert sorted(W.proposal_weights_from_tf(tfv)) == want
    rw = synth.refinement_weights(0, 1)
    tv = W.refinement_weights_to_tf(rw)
    want = sorted(W.refinement_weights_from_tf(tv))
    tv.update({"global_step": np.array(5, np.int64),
               "xception_65/entry_flow/conv1_1/weights/Adam": np.zeros((3, 3, 4, 32), np.float32),
               "xception_65/entry_flow/conv1_1/weights/Adam_1": np.zeros((3, 3, 4, 32), np.float32),
               "xception_65/entry_flow/conv1_1/BatchNorm/beta/Adam": np.zeros((32,), np.float32), "beta1_power": np.array(0.9, np.float32)})
    assert sorted(W.refinement_weights_from_tf(tv)) == want


def test_reader_on_a_bundle_assembled_by_an_independent_writer():
    """tests/golden/bundle/ was hand-assembled from the published format by tools/make_golden_bundle.py, which shares no code with
    premvos_amd/weights.py (own varints, bit-by-bit CRC-32C, own table builder with restart interval 16, shortened index
    separators, ordered-code slice keys): two data shards, prefix-compressed keys over several blocks, an int64 scalar, an int32
    matrix, a zero-size dimension and ONE PARTITIONED variable (64 + 6 rows; the 64 takes the two-byte signed ordered code)."""
    import json
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bundle")
    got = W.load_tf_checkpoint(os.path.join(root, "golden"))
    exp = json.load(open(os.path.join(root, "expected.json")))
    assert set(got) == set(exp) and "fastrcnn/partitioned/W" in got
    for k, v in exp.items():
        a = np.array(v["values"], dtype=v["dtype"]).reshape(v["shape"])
        assert got[k].dtype == a.dtype and got[k].shape == a.shape and np.array_equal(got[k], a), k
    # the slice keys the reader derives are the ones the independent writer stored
    table = W.read_table(os.path.join(root, "golden.index"))
    assert W._slice_key(b"fastrcnn/partitioned/W", [(64, 6), (0, -1)]) in table
    assert W._slice_key(b"fastrcnn/partitioned/W", [(0, 64), (0, -1)]) in table
    # ordered code, longer forms (ordered_code.cc): one byte for [-64, 64), then 7n - 1 magnitude bits in n bytes
    assert [W._oc_signed(v).hex() for v in (0, -1, 63, -64, 64, -65, 8191, 8192)] == \
        ["80", "7f", "bf", "40", "c040", "3fbf", "dfff", "e02000"]
