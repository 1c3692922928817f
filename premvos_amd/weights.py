"""Weight import without TensorFlow: a pure-Python reader (and writer) of the TF "tensor bundle" checkpoint format plus
the variable-name / layout maps of the two TF nets.

The reference restores its checkpoints through TF itself -- tensorpack ``get_model_loader(path)`` (proposal_net/train.py:655)
and ``tf.train.Saver.restore`` (refinement_net/core/Saver.py:33-48); ``simple_run.sh:31-32,39-40`` and
``refinement_net/configs/run:9`` pass checkpoint *prefixes* (``<prefix>.index`` + ``<prefix>.data-00000-of-00001``).
TensorFlow is not installed here and must not be required on an MI355X box, so the on-disk format is read directly:

  <prefix>.index   an SSTable (LevelDB table format): data blocks of prefix-compressed (key, value) entries with restart
                   arrays, each block followed by a 1-byte compression tag and a 4-byte masked CRC32C; an index block
                   mapping separator keys to block handles; a 48-byte footer (metaindex handle, index handle, padding,
                   magic 0xdb4775248b80fb57).  Key "" -> BundleHeaderProto, every other key = variable name ->
                   BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}.
  <prefix>.data-XXXXX-of-YYYYY   the raw little-endian tensor bytes.

PARITY NOTE: no TF-written checkpoint exists in the build container (simple_run.sh downloads the 3 GB weights at run
time), so the reader is validated against this module's own writer (round trip, block/restart/multi-block paths,
snappy-compressed blocks) and against the format constants above -- not yet against a TensorFlow-produced file.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           19: np.float16}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


# --------------------------------------------------------------------------------------------------
# varints / protobuf wire helpers
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        if pos >= len(buf) or shift > 63:
            raise ValueError("tensor bundle: truncated or malformed varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not (b & 0x80):
            return out, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v, pos = buf[pos:pos + n], pos + n
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield field, wt, v


def _parse_shape(buf: bytes) -> List[int]:
    dims = []
    for f, _, v in _pb_fields(buf):
        if f == 2:                                   # TensorShapeProto.Dim
            size = 0
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return dims


def _parse_entry(buf: bytes) -> dict:
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": []}
    for f, _, v in _pb_fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            e["shape"] = _parse_shape(v)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif f == 7:                                 # TensorSliceProto: repeated Extent {start = 1, length = 2 (absent = all)}
            ext = []
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    start, length = 0, -1
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1:
                            start = v3
                        elif f3 == 2:
                            length = v3
                    ext.append((start, length))
            e["slices"].append(ext)
    return e


def _oc_signed(v: int) -> bytes:
    """OrderedCode::WriteSignedNumIncreasing (tensorflow/core/lib/strings/ordered_code.cc): n bytes hold 7n - 1 magnitude bits
    behind a unary length header; one byte 0x80 ^ v for -64 <= v < 64."""
    x = ~v if v < 0 else v
    n = 1
    while x >> (7 * n - 1):
        n += 1
    b = bytearray((v & ((1 << 80) - 1)).to_bytes(10, "big")[10 - n:])
    hdr = (0xFF00 >> n) & 0xFF if n <= 8 else 0xFF
    b[0] ^= hdr
    if n == 9:
        b[1] ^= 0x80
    elif n == 10:
        b[1] ^= 0xC0
    return bytes(b)


def _slice_key(name: bytes, extents) -> bytes:
    """checkpoint::EncodeTensorNameSlice: the table key under which one partition of a partitioned variable is stored."""
    esc = b"".join(b"\x00\xff" if c == 0 else b"\xff\x00" if c == 255 else bytes([c]) for c in name)
    nd = len(extents)
    key = b"\x00" + esc + b"\x00\x01" + (bytes([1, nd]) if nd else b"\x00")
    for start, length in extents:
        key += _oc_signed(start) + _oc_signed(length)
    return key


# --------------------------------------------------------------------------------------------------
# CRC32C (Castagnoli) + LevelDB masking; snappy block decompression
_CRC_TABLE: Optional[List[int]] = None


def crc32c(data: bytes) -> int:
    global _CRC_TABLE
    if len(data) > 4096:                             # large tensors: the C helper of libpremvos_hip.so (host code)
        try:
            from . import _lib
            buf = bytes(data)
            return int(_lib.load().premvos_crc32c_host(buf, len(buf)))
        except Exception:
            pass
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask_crc(c: int) -> int:
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def snappy_decompress(buf: bytes) -> bytes:
    n, pos = _varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        for _ in range(ln):                          # overlapping copies are legal
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy: length mismatch")
    return bytes(out)


# --------------------------------------------------------------------------------------------------
# SSTable
def _read_block(data: bytes, offset: int, size: int, verify: bool = True) -> bytes:
    if offset < 0 or size < 4 or offset + size + 5 > len(data):          # block + 1-byte type + 4-byte crc must lie inside the file
        raise ValueError(f"tensor bundle index: block handle ({offset}, {size}) points outside the {len(data)}-byte file (truncated?)")
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        stored = struct.unpack("<I", data[offset + size + 1:offset + size + 5])[0]
        if _mask_crc(crc32c(raw + bytes([ctype]))) != stored:
            raise ValueError("tensor bundle index: block checksum mismatch")
    if ctype == 0:
        return raw
    if ctype == 1:
        return snappy_decompress(raw)
    raise ValueError(f"unknown block compression {ctype}")


def _block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    num_restarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path: str, verify: bool = True) -> Dict[bytes, bytes]:
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != TABLE_MAGIC:
        raise ValueError(f"{path}: not a TensorFlow tensor-bundle index (bad magic)")
    footer = data[-48:]
    pos = 0
    _, pos = _varint(footer, pos)                    # metaindex handle
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)                 # index handle
    isize, pos = _varint(footer, pos)
    out: Dict[bytes, bytes] = {}
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = _varint(handle, 0)
        bsize, _ = _varint(handle, p)
        for k, v in _block_entries(_read_block(data, boff, bsize, verify)):
            out[k] = v
    return out


def load_tf_checkpoint(prefix: str, verify: bool = True) -> Dict[str, np.ndarray]:
    """Every variable of a TF tensor-bundle checkpoint as numpy arrays (TF layouts, TF names)."""
    table = read_table(prefix + ".index", verify)
    num_shards = 1
    for f, _, v in _pb_fields(table.get(b"", b"")):
        if f == 1:
            num_shards = v
        if f == 2 and v != 0:
            raise ValueError("big-endian bundles are not supported")
    shards: Dict[int, np.memmap] = {}
    out: Dict[str, np.ndarray] = {}

    def tensor(key: bytes, e: dict) -> np.ndarray:
        sid = e["shard_id"]
        if sid not in shards:
            fn = f"{prefix}.data-{sid:05d}-of-{num_shards:05d}"
            if not (0 <= sid < num_shards) or not os.path.exists(fn):
                raise FileNotFoundError(f"{key!r} lives in shard {sid} of {num_shards}, but {fn} does not exist")
            shards[sid] = np.memmap(fn, dtype=np.uint8, mode="r")
        if e["offset"] + e["size"] > shards[sid].shape[0]:
            raise ValueError(f"{key!r}: bytes [{e['offset']}, {e['offset'] + e['size']}) lie beyond the end of shard {sid} (truncated file?)")
        want = int(np.prod(e["shape"], dtype=np.int64)) * np.dtype(_DTYPES[e["dtype"]]).itemsize
        if want != e["size"]:
            raise ValueError(f"{key!r}: shape {e['shape']} needs {want} bytes, the entry holds {e['size']}")
        raw = np.asarray(shards[sid][e["offset"]:e["offset"] + e["size"]])
        if verify and e["crc32c"] is not None and _mask_crc(crc32c(raw.tobytes())) != e["crc32c"]:
            raise ValueError(f"{key!r}: tensor checksum mismatch")
        return raw.view(_DTYPES[e["dtype"]]).reshape(e["shape"]).copy()

    for key, val in table.items():
        if key == b"":
            continue
        e = _parse_entry(val)
        if key[:1] == b"\x00":
            continue                                 # an ordered-code slice key: read through its variable's entry below
        if e["dtype"] not in _DTYPES:
            continue                                 # strings etc. (global_step is int64 and is kept)
        if e["slices"]:
            # a variable saved through a partitioner (neither tensorpack, proposal_net/train.py:655, nor the slim graph, Saver.py:
            # 33-48, of the reference defines one -- but tf.train.Saver writes them this way whenever a graph does): the full-name
            # entry has the shape and the slice list, every partition is an entry of its own under an ordered-code key
            full = np.zeros(e["shape"], dtype=_DTYPES[e["dtype"]])
            covered = 0
            for ext in e["slices"]:
                sk = _slice_key(key, ext)
                if sk not in table:
                    raise ValueError(f"{key!r}: partition {ext} is listed in the variable's entry but has no slice entry in the index")
                se = _parse_entry(table[sk])
                idx = tuple(slice(st, None if ln < 0 else st + ln) for st, ln in ext)
                part = tensor(key + b" " + repr(ext).encode(), se)
                if full[idx].shape != part.shape:
                    raise ValueError(f"{key!r}: partition {ext} has shape {list(part.shape)}, the variable's shape {e['shape']} needs "
                                     f"{list(full[idx].shape)}")
                full[idx] = part
                covered += part.size
            if covered != full.size:
                raise ValueError(f"{key!r}: its partitions cover {covered} of {full.size} elements")
            out[key.decode()] = full
            continue
        out[key.decode()] = tensor(key, e)
    return out


def save_tf_checkpoint(prefix: str, variables: Dict[str, np.ndarray], block_size: int = 4096,
                       restart_interval: int = 16, num_shards: int = 1) -> None:
    """Writer of the same format (no compression): used to export weights for the reference and to test the reader;
    mirrors tensorflow BundleWriter + table::TableBuilder.  ``num_shards`` > 1 lays the variables out round-robin over
    ``<prefix>.data-0000k-of-0000N`` files the way a merged multi-device save does (BundleEntryProto.shard_id, per-shard
    offsets, BundleHeaderProto.num_shards)."""
    names = sorted(variables, key=lambda s: s.encode())
    data = [bytearray() for _ in range(num_shards)]
    entries: List[Tuple[bytes, bytes]] = [(b"", b"\x08" + _put_varint(num_shards) + b"\x10\x00\x1a\x02\x08\x01")]   # LITTLE, version{producer=1}
    for i, n in enumerate(names):
        a = np.asarray(variables[n], order="C")
        raw = a.tobytes()
        sid = i % num_shards
        shape = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in a.shape))
        e = (b"\x08" + _put_varint(_DTYPE_IDS[a.dtype]) + b"\x12" + _put_varint(len(shape)) + shape + b"\x18" + _put_varint(sid) +
             b"\x20" + _put_varint(len(data[sid])) + b"\x28" + _put_varint(len(raw)) + b"\x35" +
             struct.pack("<I", _mask_crc(crc32c(raw))))
        entries.append((n.encode(), e))
        data[sid] += raw
    for sid in range(num_shards):
        with open(f"{prefix}.data-{sid:05d}-of-{num_shards:05d}", "wb") as f:
            f.write(bytes(data[sid]))

    out = bytearray()

    def emit_block(items: List[Tuple[bytes, bytes]]) -> bytes:
        buf, restarts, last = bytearray(), [], b""
        for i, (k, v) in enumerate(items):
            shared = 0
            if i % restart_interval == 0:
                restarts.append(len(buf))
            else:
                while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                    shared += 1
            buf += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
            last = k
        for r in restarts or [0]:
            buf += struct.pack("<I", r)
        buf += struct.pack("<I", max(len(restarts), 1))
        off = len(out)
        out.extend(buf)
        out.extend(b"\x00" + struct.pack("<I", _mask_crc(crc32c(bytes(buf) + b"\x00"))))
        return _put_varint(off) + _put_varint(len(buf))

    index: List[Tuple[bytes, bytes]] = []
    cur: List[Tuple[bytes, bytes]] = []
    size = 0
    for k, v in entries:
        cur.append((k, v))
        size += len(k) + len(v) + 6
        if size >= block_size:
            index.append((cur[-1][0], emit_block(cur)))
            cur, size = [], 0
    if cur:
        index.append((cur[-1][0], emit_block(cur)))
    meta = emit_block([])
    idx = emit_block(index)
    footer = meta + idx
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))


# --------------------------------------------------------------------------------------------------
# name / layout maps
def _t(x: np.ndarray):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x))


def proposal_weights_from_tf(v: Dict[str, np.ndarray]) -> Dict[str, object]:
    """tensorpack variable names (basemodel.py:31-46,53-59,80-98; model.py:39-42,388-393,507-508,562-564) -> the dict
    ``ProposalNet`` takes.  Conv kernels HWIO -> OIHW, FC [in,out] -> [out,in], Deconv2D filter [kh,kw,out,in] ->
    [in,out,kh,kw]; BatchNorm {gamma,beta,mean/EMA,variance/EMA} -> {gamma,beta,mean,var}."""
    w: Dict[str, object] = {}
    for name, a in v.items():
        if name.endswith("/bn/gamma"):
            p = name[:-len("/gamma")]
            w[p] = {"gamma": _t(v[p + "/gamma"]), "beta": _t(v[p + "/beta"]), "mean": _t(v[p + "/mean/EMA"]),
                    "var": _t(v[p + "/variance/EMA"])}
        elif name.endswith("/W"):
            if name.startswith(("fastrcnn/", "secondclassification/")):
                w[name] = _t(a.T)
            elif name == "maskrcnn/deconv/W":
                w[name] = _t(a.transpose(3, 2, 0, 1))
            else:
                w[name] = _t(a.transpose(3, 2, 0, 1))
        elif name.endswith("/b"):
            w[name] = _t(a)
    return w


def proposal_weights_to_tf(w: Dict[str, object]) -> Dict[str, np.ndarray]:
    v: Dict[str, np.ndarray] = {}
    for name, a in w.items():
        if isinstance(a, dict):
            v[name + "/gamma"], v[name + "/beta"] = a["gamma"].numpy(), a["beta"].numpy()
            v[name + "/mean/EMA"], v[name + "/variance/EMA"] = a["mean"].numpy(), a["var"].numpy()
        elif name.endswith("/W"):
            x = a.numpy()
            if name.startswith(("fastrcnn/", "secondclassification/")):
                v[name] = np.ascontiguousarray(x.T)
            elif name == "maskrcnn/deconv/W":
                v[name] = np.ascontiguousarray(x.transpose(2, 3, 1, 0))
            else:
                v[name] = np.ascontiguousarray(x.transpose(2, 3, 1, 0))
        else:
            v[name] = a.numpy()
    return v


def refinement_weights_from_tf(v: Dict[str, np.ndarray]) -> Dict[str, object]:
    """slim variable names (network/deeplab/model.py:61-66, core/xception.py:172-177,248,275,282,430-433,505-550;
    Saver.py:103-106) -> the dict ``RefinementNet`` takes.  The backbone prefix 'xception_65/' is stripped;
    weights HWIO -> OIHW, depthwise_weights [3,3,C,1] -> [C,1,3,3]; BatchNorm moving_mean/moving_variance."""
    w: Dict[str, object] = {}
    for name, a in v.items():
        key = name[len("xception_65/"):] if name.startswith("xception_65/") else name
        if key.endswith("/BatchNorm/gamma"):
            p = name[:-len("/gamma")]
            w[key[:-len("/gamma")]] = {"gamma": _t(v[p + "/gamma"]), "beta": _t(v[p + "/beta"]),
                                       "mean": _t(v[p + "/moving_mean"]), "var": _t(v[p + "/moving_variance"])}
        elif key.endswith("/depthwise_weights"):
            w[key] = _t(a.transpose(2, 3, 0, 1))
        elif key.endswith("/weights"):
            w[key] = _t(a.transpose(3, 2, 0, 1))
        elif key.endswith("/biases"):
            w[key] = _t(a)
    return w


def refinement_weights_to_tf(w: Dict[str, object]) -> Dict[str, np.ndarray]:
    v: Dict[str, np.ndarray] = {}
    head = ("image_pooling", "aspp", "concat_projection", "decoder/", "logits/")
    for key, a in w.items():
        name = key if key.startswith(head) else "xception_65/" + key
        if isinstance(a, dict):
            v[name + "/gamma"], v[name + "/beta"] = a["gamma"].numpy(), a["beta"].numpy()
            v[name + "/moving_mean"], v[name + "/moving_variance"] = a["mean"].numpy(), a["var"].numpy()
        elif key.endswith("/depthwise_weights"):
            v[name] = np.ascontiguousarray(a.numpy().transpose(2, 3, 0, 1))
        elif key.endswith("/weights"):
            v[name] = np.ascontiguousarray(a.numpy().transpose(2, 3, 1, 0))
        else:
            v[name] = a.numpy()
    return v


def reid_weights_from_tf(v: Dict[str, np.ndarray]) -> Dict[str, object]:
    """ReID_net variable names (network/NetworkLayers.py:62-76 'W' / 'b'; Util_Network.py:87-97
    '<scope>/{beta,gamma,mean_ema,var_ema}') -> the dict ``ReIDNet`` takes: conv W HWIO -> OIHW, FC W [in,out] -> [out,in]."""
    w: Dict[str, object] = {}
    for name, a in v.items():
        if name.endswith("/gamma"):
            p = name[:-len("/gamma")]
            w[p] = {"gamma": _t(v[p + "/gamma"]), "beta": _t(v[p + "/beta"]), "mean": _t(v[p + "/mean_ema"]),
                    "var": _t(v[p + "/var_ema"])}
        elif name.rsplit("/", 1)[-1][:1] == "W" and a.ndim == 4:
            w[name] = _t(a.transpose(3, 2, 0, 1))
        elif name.endswith("/W") and a.ndim == 2:
            w[name] = _t(a.T)
        elif name.endswith("/b"):
            w[name] = _t(a)
    return w


def reid_weights_to_tf(w: Dict[str, object]) -> Dict[str, np.ndarray]:
    v: Dict[str, np.ndarray] = {}
    for key, a in w.items():
        if isinstance(a, dict):
            v[key + "/gamma"], v[key + "/beta"] = a["gamma"].numpy(), a["beta"].numpy()
            v[key + "/mean_ema"], v[key + "/var_ema"] = a["mean"].numpy(), a["var"].numpy()
        elif a.dim() == 4:
            v[key] = np.ascontiguousarray(a.numpy().transpose(2, 3, 1, 0))
        elif a.dim() == 2:
            v[key] = np.ascontiguousarray(a.numpy().T)
        else:
            v[key] = a.numpy()
    return v


def expected_keys(kind: str, weights: Dict[str, object]) -> List[str]:
    """The names the net of ``kind`` packs, for the depth read off ``weights`` (product tables: RESNET layout of
    proposal/model.py, module_plan of refinement/model.py); the names themselves are pinned by the reference's graph code
    (tests/golden/{proposal,deeplab}_host_refs.json: every variable the graph requests)."""
    keys: List[str] = []
    if kind == "proposal":
        blocks = tuple(len({k.split("/")[1] for k in weights if k.startswith(f"group{g}/block")}) for g in range(4))
        keys += ["conv0/W", "conv0/bn"]
        cin = 64
        for g, (feat, cnt) in enumerate(zip((64, 128, 256, 512), blocks)):
            for i in range(cnt):
                p = f"group{g}/block{i}"
                for c in ("conv1", "conv2", "conv3"):
                    keys += [f"{p}/{c}/W", f"{p}/{c}/bn"]
                if cin != 4 * feat:
                    keys += [f"{p}/convshortcut/W", f"{p}/convshortcut/bn"]
                cin = 4 * feat
        for h in ("rpn/conv0", "rpn/class", "rpn/box", "fastrcnn/class", "fastrcnn/box", "secondclassification/class"):
            keys += [h + "/W", h + "/b"]
    elif kind == "refinement":
        from .refinement.model import module_plan
        nm = len({k.split("/")[2] for k in weights if k.startswith("middle_flow/block1/unit_")})
        def conv(scope):
            return [scope + "/weights", scope + "/BatchNorm"]
        def dw(scope):
            return [scope + "/depthwise_weights", scope + "/BatchNorm"]
        keys += conv("entry_flow/conv1_1") + conv("entry_flow/conv1_2")
        for prefix, _, _, skip, _, _, _ in module_plan(nm):
            for i in (1, 2, 3):
                keys += dw(f"{prefix}/separable_conv{i}_depthwise") + conv(f"{prefix}/separable_conv{i}_pointwise")
            if skip == "conv":
                keys += conv(prefix + "/shortcut")
        keys += conv("image_pooling") + conv("aspp0") + conv("concat_projection") + conv("decoder/feature_projection0")
        for i in (1, 2, 3):
            keys += dw(f"aspp{i}_depthwise") + conv(f"aspp{i}_pointwise")
        for j in (0, 1):
            keys += dw(f"decoder/decoder_conv{j}_depthwise") + conv(f"decoder/decoder_conv{j}_pointwise")
        keys += ["logits/features/weights", "logits/features/biases"]
    return keys


def load_any(path: str, kind: str) -> Dict[str, object]:
    """``path`` = a TF checkpoint prefix (what simple_run.sh passes: ``<path>.index`` + ``<path>.data-*``, checksums verified)
    or a torch pickle of the name->tensor dict.  Optimizer slots / global_step in a TF checkpoint are ignored by the name
    maps; a weight the net needs that the file does not hold is an error that lists the missing names."""
    if os.path.exists(path + ".index"):
        v = load_tf_checkpoint(path)
        w = {"proposal": proposal_weights_from_tf, "refinement": refinement_weights_from_tf,
             "reid": reid_weights_from_tf}[kind](v)
    elif os.path.exists(path):
        import torch
        w = torch.load(path, map_location="cpu")
        if not isinstance(w, dict):
            raise ValueError(f"{path}: expected a pickled dict of name -> tensor, got {type(w).__name__}")
    else:
        raise FileNotFoundError(f"no checkpoint at {path!r}: neither the TF bundle {path}.index / {path}.data-* nor a torch "
                                f"pickle of that name exists")
    missing = [k for k in expected_keys(kind, w) if k not in w]
    if missing:
        raise KeyError(f"{path}: {len(missing)} weight(s) the {kind} net needs are missing, e.g. {missing[:8]}")
    return w
