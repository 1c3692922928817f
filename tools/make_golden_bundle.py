#!/usr/bin/env python
"""Hand-assembles a TensorFlow "tensor bundle" checkpoint from the PUBLISHED on-disk format, byte by byte, with code that
shares nothing with premvos_amd/weights.py -- a second author path for the reader's golden file:

    tests/golden/bundle/golden.index, golden.data-00000-of-00002, golden.data-00001-of-00002   the bundle
    tests/golden/bundle/expected.json                                                          what a correct reader returns

What the file exercises that premvos_amd.weights' own writer is not the only witness of: a LevelDB table with restart
interval 16 and real prefix compression across entries, several data blocks with SHORTENED index separators (the index key of a
block is not its last key), two data shards, a non-float variable, a scalar, a zero-size dimension, AND one variable saved
through a partitioner: its full-name entry carries `slices` and no data, the partitions live under ordered-code slice keys
(tensorflow/core/util/tensor_slice_util / saved_tensor_slice_util: EncodeTensorNameSlice).

Format sources (public): LevelDB doc/table_format.md + block_builder.cc; tensorflow/core/protobuf/tensor_bundle.proto;
tensorflow/core/lib/hash/crc32c.h (mask = rotr15 + 0xa282ead8); tensorflow/core/lib/strings/ordered_code.cc.  No block is
compressed (TF writes bundles uncompressed).  Deterministic: running it again gives the same bytes.
"""
import json
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "bundle")


# ---- CRC-32C (Castagnoli), bit by bit (slow and obviously right) -------------------------------------------------------
def crc32c(b: bytes) -> int:
    c = 0xFFFFFFFF
    for x in b:
        c ^= x
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def masked(c: int) -> int:
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire format ------------------------------------------------------------------------------------------------
def vint(v: int) -> bytes:
    v &= (1 << 64) - 1
    o = bytearray()
    while True:
        if v < 0x80:
            o.append(v)
            return bytes(o)
        o.append((v & 0x7F) | 0x80)
        v >>= 7


def f_varint(field, v):
    return vint(field << 3) + vint(v)


def f_bytes(field, b):
    return vint((field << 3) | 2) + vint(len(b)) + b


def f_fixed32(field, v):
    return vint((field << 3) | 5) + struct.pack("<I", v)


def shape_proto(shape):                       # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
    return b"".join(f_bytes(2, f_varint(1, d)) for d in shape)


def slice_proto(extents):                     # TensorSliceProto: repeated Extent extent = 1 { int64 start = 1; oneof { int64 length = 2 } }
    out = b""
    for start, length in extents:
        e = (f_varint(1, start) if start else b"") + (f_varint(2, length) if length >= 0 else b"")   # full extent: no length
        out += f_bytes(1, e)
    return out


DT = {"float32": 1, "int32": 3, "int64": 9}


def entry_proto(dtype, shape, shard, offset, size, crc, slices=()):
    e = f_varint(1, DT[dtype]) + f_bytes(2, shape_proto(shape))
    if shard:
        e += f_varint(3, shard)
    if offset:
        e += f_varint(4, offset)
    if size:
        e += f_varint(5, size)
    if crc is not None:
        e += f_fixed32(6, crc)
    for s in slices:
        e += f_bytes(7, slice_proto(s))
    return e


# ---- ordered code (slice keys) ---------------------------------------------------------------------------------------------
def oc_num_increasing(v: int) -> bytes:
    body = b"" if v == 0 else v.to_bytes((v.bit_length() + 7) // 8, "big")
    return bytes([len(body)]) + body


def oc_string(s: bytes) -> bytes:
    return b"".join(b"\x00\xff" if c == 0 else b"\xff\x00" if c == 255 else bytes([c]) for c in s) + b"\x00\x01"


def oc_signed_increasing(v: int) -> bytes:
    x = ~v if v < 0 else v
    if x < 64:
        return bytes([(0x80 ^ v) & 0xFF])
    n = 1
    while x >= (1 << (7 * n - 1)):            # n bytes carry 7n - 1 magnitude bits (n <= 8)
        n += 1
    buf = bytearray((v & ((1 << 80) - 1)).to_bytes(10, "big"))      # sign-extended big endian
    head = {2: (0xC0, 0), 3: (0xE0, 0), 4: (0xF0, 0), 5: (0xF8, 0), 6: (0xFC, 0), 7: (0xFE, 0), 8: (0xFF, 0)}[n]
    b = buf[10 - n:]
    b[0] ^= head[0]
    b[1] ^= head[1]
    return bytes(b)


def slice_key(name: bytes, extents) -> bytes:
    k = oc_num_increasing(0) + oc_string(name) + oc_num_increasing(len(extents))
    for start, length in extents:
        k += oc_signed_increasing(start) + oc_signed_increasing(length)
    return k


# ---- LevelDB table -----------------------------------------------------------------------------------------------------------
def build_block(items, restart_interval):
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += vint(shared) + vint(len(k) - shared) + vint(len(v)) + k[shared:] + v
        last = k
    for r in restarts or [0]:               # (BlockBuilder starts with one restart point at offset 0, also in an empty block)
        out += struct.pack("<I", r)
    out += struct.pack("<I", max(len(restarts), 1))
    return bytes(out)


def shortest_separator(a: bytes, b: bytes) -> bytes:
    """leveldb BytewiseComparator::FindShortestSeparator: a key >= a and < b, as short as it gets."""
    n = 0
    while n < min(len(a), len(b)) and a[n] == b[n]:
        n += 1
    if n < min(len(a), len(b)) and a[n] < 0xFF and a[n] + 1 < b[n]:
        return a[:n] + bytes([a[n] + 1])
    return a


def short_successor(a: bytes) -> bytes:
    for i, c in enumerate(a):
        if c != 0xFF:
            return a[:i] + bytes([c + 1])
    return a


def build_table(items, per_block, restart_interval=16):
    items = sorted(items)
    file, index = bytearray(), []

    def emit(block):
        off = len(file)
        file.extend(block)
        file.extend(b"\x00" + struct.pack("<I", masked(crc32c(block + b"\x00"))))
        return vint(off) + vint(len(block))
    blocks = [items[i:i + per_block] for i in range(0, len(items), per_block)]
    for bi, blk in enumerate(blocks):
        handle = emit(build_block(blk, restart_interval))
        sep = shortest_separator(blk[-1][0], blocks[bi + 1][0][0]) if bi + 1 < len(blocks) else short_successor(blk[-1][0])
        index.append((sep, handle))
    meta = emit(build_block([], 1))
    idx = emit(build_block(index, 1))
    footer = meta + idx
    file.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57))
    return bytes(file)


def main():
    rng = np.random.default_rng(20260928)
    names = [f"group2/block{i}/conv{j}/W" for i in range(6) for j in (1, 2, 3)] + \
            [f"group2/block{i}/conv{j}/bn/{s}" for i in range(2) for j in (1, 2) for s in ("gamma", "beta", "mean/EMA", "variance/EMA")]
    variables = {n: rng.standard_normal((1, 1, 3 + k % 4, 5)).astype(np.float32) if n.endswith("/W") else
                 rng.standard_normal((5,)).astype(np.float32) for k, n in enumerate(names)}
    variables["global_step"] = np.array(123456789012, np.int64)                      # a scalar, int64
    variables["xception_65/entry_flow/conv1_1/weights"] = rng.standard_normal((3, 3, 4, 8)).astype(np.float32)
    variables["empty/dim"] = np.zeros((0, 7), np.float32)
    variables["labels"] = rng.integers(-5, 5, (4, 3)).astype(np.int32)
    part = rng.standard_normal((70, 6)).astype(np.float32)                           # saved as two row partitions: 64 + 6 rows (64 needs the two-byte signed ordered code)
    shards = [bytearray(), bytearray()]
    items = [(b"", f_varint(1, 2) + f_varint(2, 0) + f_bytes(3, f_varint(1, 1)))]     # BundleHeaderProto: 2 shards, little, version{producer 1}
    for k, (name, arr) in enumerate(sorted(variables.items())):
        sid = k % 2
        raw = arr.tobytes()
        off = len(shards[sid])
        shards[sid] += raw
        items.append((name.encode(), entry_proto(str(arr.dtype), arr.shape, sid, off, len(raw), masked(crc32c(raw)))))
    pname = b"fastrcnn/partitioned/W"
    cuts = [((0, 64), (0, -1)), ((64, 6), (0, -1))]                                   # rows [0, 64) and [64, 70), all columns
    items.append((pname, entry_proto("float32", part.shape, 0, 0, 0, None, slices=cuts)))
    for (r0, rn), _ in cuts:
        raw = part[r0:r0 + rn].tobytes()
        off = len(shards[1])
        shards[1] += raw
        items.append((slice_key(pname, [(r0, rn), (0, -1)]),
                      entry_proto("float32", (rn, 6), 1, off, len(raw), masked(crc32c(raw)))))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "golden.index"), "wb") as f:
        f.write(build_table(items, per_block=9))
    for sid, data in enumerate(shards):
        with open(os.path.join(OUT, f"golden.data-{sid:05d}-of-00002"), "wb") as f:
            f.write(bytes(data))
    expected = {n: {"dtype": str(a.dtype), "shape": list(a.shape), "values": a.reshape(-1).tolist()} for n, a in variables.items()}
    expected[pname.decode()] = {"dtype": "float32", "shape": [70, 6], "values": part.reshape(-1).tolist()}
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump(expected, f)
    print("wrote", OUT, {fn: os.path.getsize(os.path.join(OUT, fn)) for fn in sorted(os.listdir(OUT))})


if __name__ == "__main__":
    main()
