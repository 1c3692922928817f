"""Optional GPU JPEG decode, the parts that need no GPU:
  * oracle/jpeg_oracle.py pinned against the library the reference's readers use (libjpeg-turbo inside PIL): same bytes;
  * the host half of the C-ABI (premvos_jpeg_entropy_decode_host: marker parsing + Huffman decoding) against the oracle's
    coefficients, its refusals (PREMVOS_EUNSUPPORTED) and its error paths."""
import ctypes as C
import io

import numpy as np
import pytest
from PIL import Image

from oracle import jpeg_oracle as jo


def picture(h, w, seed=0, grey=False):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(x / 7.0 + y / 11.0), 128 + 90 * np.cos(x / 5.0 - y / 13.0), 60 + 0.5 * x + 0.3 * y], -1)
    im = np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)
    return im[..., 0] if grey else im


def jpeg_bytes(im, **kw):
    b = io.BytesIO()
    Image.fromarray(im).save(b, "JPEG", **kw)
    return b.getvalue()


def with_luma_sampling(data, byte):
    """The same file with another luma sampling byte in its SOF0 segment (PIL cannot write 4:1:1 / 4:4:0; the header is enough)."""
    i = data.index(b"\xff\xc0")
    assert data[i + 11] == 0x22                   # component 1: id, sampling, table
    return data[:i + 11] + bytes([byte]) + data[i + 12:]


def pil_rgb(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


CASES = [(48, 64, dict(quality=75, subsampling=2)), (37, 53, dict(quality=90, subsampling=2)), (37, 53, dict(quality=90, subsampling=1)),
         (37, 53, dict(quality=100, subsampling=0)), (40, 71, dict(quality=60, subsampling=2, optimize=True)),
         (33, 47, dict(quality=85, subsampling=2, restart_marker_blocks=3)), (33, 47, dict(quality=85, subsampling=1, restart_marker_rows=1)),
         (16, 16, dict(quality=30, subsampling=2)), (9, 7, dict(quality=95, subsampling=2)), (8, 5, dict(quality=95, subsampling=0)),
         (97, 131, dict(quality=5, subsampling=2)), (120, 214, dict(quality=95, subsampling=2))]


@pytest.mark.parametrize("h,w,kw", CASES)
def test_oracle_returns_the_bytes_of_the_library(h, w, kw):
    data = jpeg_bytes(picture(h, w, seed=h + w), **kw)
    assert np.array_equal(jo.decode(data), pil_rgb(data))


def test_oracle_grey_and_refusals():
    data = jpeg_bytes(picture(30, 41, grey=True), quality=80)
    assert np.array_equal(jo.decode(data), pil_rgb(data))
    with pytest.raises(jo.Unsupported):
        jo.decode(jpeg_bytes(picture(30, 41), quality=80, progressive=True))
    for byte in (0x41, 0x12):                                  # 4:1:1, 4:4:0
        with pytest.raises(jo.Unsupported):
            jo.decode(with_luma_sampling(jpeg_bytes(picture(30, 41), quality=80, subsampling=2), byte))
    with pytest.raises(jo.Unsupported):                        # chroma plane of two columns: the library replicates, not filters
        jo.decode(jpeg_bytes(picture(12, 4), quality=80, subsampling=2))


def test_info_struct_layout_matches_the_header():
    from premvos_amd import jpeg
    assert C.sizeof(jpeg.JpegInfo) == 14 * 4 + 4 * 8 + 3 * 64 * 2
    assert jpeg.JpegInfo.coef_offset.offset == 56 and jpeg.JpegInfo.quant.offset == 88


@pytest.mark.parametrize("h,w,kw", CASES)
def test_host_entropy_decoder_matches_the_oracle(h, w, kw):
    from premvos_amd import jpeg
    data = jpeg_bytes(picture(h, w, seed=h + w), **kw)
    f = jo.parse(data)
    ref = jo.entropy_decode(f)
    d = jpeg.entropy_decode(data)
    info = d.info
    assert (info.height, info.width, info.ncomp) == (h, w, 3) and d.shape == (h, w, 3)
    assert (info.hs, info.vs) == f["comps"][0][1:3]
    for c in range(3):
        n = info.blocks_w[c] * info.blocks_h[c] * 64
        got = d.coef[info.coef_offset[c]:info.coef_offset[c] + n].numpy().reshape(info.blocks_h[c], info.blocks_w[c], 64)
        assert np.array_equal(got, ref[c]), c
        assert np.array_equal(np.array(info.quant[c][:]), f["qt"][f["comps"][c][3]])
    assert info.coef_count == sum(r.size for r in ref)


def test_host_decoder_grey_header_only_and_16_bit_tables():
    from premvos_amd import jpeg
    data = jpeg_bytes(picture(30, 41, grey=True), quality=80)
    info = jpeg.header(data)
    assert (info.ncomp, info.hs, info.vs, info.mcux, info.mcuy) == (1, 1, 1, 6, 4)
    d = jpeg.entropy_decode(data)
    ref = jo.entropy_decode(jo.parse(data))
    assert np.array_equal(d.coef[:info.coef_count].numpy().reshape(ref[0].shape), ref[0])
    # quality 1 gives quantisation steps above 255: a 16-bit DQT segment
    data = jpeg_bytes(picture(24, 24), qtables=[[300] * 64, [700] * 64], subsampling=0)
    f = jo.parse(data)
    info = jpeg.header(data)
    assert max(info.quant[0][:]) == 300 and max(info.quant[1][:]) == 700
    assert np.array_equal(jo.decode(data), pil_rgb(data))
    d = jpeg.entropy_decode(data)
    assert np.array_equal(d.coef[:24 * 24].numpy().reshape(3, 3, 64), jo.entropy_decode(f)[0])


def test_host_decoder_refuses_what_it_does_not_cover_and_reports_corrupt_files():
    from premvos_amd import _lib, jpeg
    with pytest.raises(jpeg.Unsupported):
        jpeg.entropy_decode(jpeg_bytes(picture(30, 41), quality=80, progressive=True))
    for byte in (0x41, 0x12):                                  # 4:1:1, 4:4:0
        with pytest.raises(jpeg.Unsupported, match="sampling"):
            jpeg.entropy_decode(with_luma_sampling(jpeg_bytes(picture(30, 41), quality=80, subsampling=2), byte))
    with pytest.raises(jpeg.Unsupported, match="narrower"):
        jpeg.entropy_decode(jpeg_bytes(picture(12, 4), quality=80, subsampling=2))
    cmyk = io.BytesIO()
    Image.fromarray(picture(16, 16)).convert("CMYK").save(cmyk, "JPEG")
    with pytest.raises(jpeg.Unsupported):
        jpeg.entropy_decode(cmyk.getvalue())
    # the drivers' host half falls back to the default reader for such files (and for non-JPEG frames)
    prog = jpeg_bytes(picture(30, 41), quality=80, progressive=True)
    assert np.array_equal(jpeg.host_stage(prog), pil_rgb(prog))
    png = io.BytesIO()
    Image.fromarray(picture(10, 12)).save(png, "PNG")
    assert np.array_equal(jpeg.host_stage(png.getvalue()), picture(10, 12))
    good = jpeg_bytes(picture(64, 64), quality=90)
    with pytest.raises(_lib.PremvosError, match="ends early"):
        jpeg.entropy_decode(good[:len(good) // 2])
    with pytest.raises(_lib.PremvosError, match="SOI"):
        jpeg.entropy_decode(b"not a jpeg at all")
    with pytest.raises(_lib.PremvosError, match="truncated|runs past"):
        jpeg.entropy_decode(good[:40])
    lib = _lib.load()
    info = jpeg.JpegInfo()
    small = np.zeros(16, np.int16)
    assert lib.premvos_jpeg_entropy_decode_host(good, len(good), C.byref(info), small.ctypes.data, small.size) == -1
    assert b"coefficient buffer" in lib.premvos_last_error()


def test_host_decoder_survives_malformed_tables_huge_headers_and_random_damage():
    """an over-subscribed DHT (more codes of a length than the code space holds) used to overrun the
    512-entry fast table on the stack; SOF dimensions were unbounded.  Now: clean errors, and a seeded fuzz over truncations and
    byte flips of good files never crashes -- every outcome is a decoded frame, ``Unsupported`` or a ``PremvosError``."""
    from premvos_amd import _lib, jpeg
    # SOI + one DHT whose counts claim 255 one-bit codes (parse_header's cnt <= 256 let it through)
    counts = bytes([255] + [0] * 15)
    seg = bytes([0x00]) + counts + bytes(255)
    bad = b"\xff\xd8\xff\xc4" + (len(seg) + 2).to_bytes(2, "big") + seg
    with pytest.raises(_lib.PremvosError, match="prefix code"):
        jpeg.header(bad)
    for counts in ([0, 5] + [0] * 14, [2, 1] + [0] * 14, [1, 1, 1, 1, 1, 1, 1, 1, 255] + [0] * 7):      # 5 two-bit codes; 2 + 1 > 4 ...
        n = sum(counts)
        seg = bytes([0x10]) + bytes(counts) + bytes(range(n % 256)) * 1 + bytes(max(0, n - n % 256))
        seg = seg[:17 + n]
        data = b"\xff\xd8\xff\xc4" + (len(seg) + 2).to_bytes(2, "big") + seg
        with pytest.raises(_lib.PremvosError, match="prefix code|bad DHT"):
            jpeg.header(data)
    # a complete (exactly full) code is fine: 2 one-bit codes
    good = jpeg_bytes(picture(64, 64), quality=90)
    assert jpeg.header(good).width == 64
    # a header claiming 65535 x 65535 pixels must not make the caller allocate 8 GB
    sof = good.index(b"\xff\xc0")
    huge = bytearray(good)
    huge[sof + 5:sof + 9] = b"\xff\xff\xff\xff"
    with pytest.raises(_lib.PremvosError, match="64 Mpixel"):
        jpeg.header(bytes(huge))
    # fuzz: truncations + byte flips of three files (4:2:0 with optimised tables, 4:4:4, grey)
    rng = np.random.default_rng(7)
    files = [jpeg_bytes(picture(40, 56), quality=70, subsampling=2, optimize=True), jpeg_bytes(picture(33, 47), quality=95, subsampling=0),
             jpeg_bytes(picture(24, 24, grey=True), quality=85)]
    outcomes = {"ok": 0, "unsupported": 0, "error": 0}
    for f in files:
        for trial in range(150):
            b = bytearray(f)
            if trial % 3 == 0:
                b = b[:int(rng.integers(2, len(b)))]
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
            try:
                d = jpeg.entropy_decode(bytes(b))
                assert d.coef.numel() >= d.info.coef_count
                outcomes["ok"] += 1
            except jpeg.Unsupported:
                outcomes["unsupported"] += 1
            except _lib.PremvosError:
                outcomes["error"] += 1
    assert sum(outcomes.values()) == 450 and outcomes["error"] > 50 and outcomes["ok"] > 10, outcomes


def test_loader_is_the_default_reader_unless_enabled(monkeypatch):
    from premvos_amd import jpeg
    monkeypatch.delenv("PREMVOS_GPU_JPEG", raising=False)
    assert jpeg.loader() is jpeg._pil_rgb
    monkeypatch.setenv("PREMVOS_GPU_JPEG", "1")
    assert jpeg.loader() is jpeg.host_stage


def test_frame_helpers_on_host_arrays_and_tensors():
    """``stack_frames`` / ``to_device`` (what every driver calls at its upload site) with the default reader's arrays: one [n,H,W,3]
    uint8 tensor, RGB kept or flipped to the cv2.imread order; an RGBA array loses its alpha plane; a tensor passes through."""
    import torch
    from premvos_amd import jpeg
    a, b = picture(6, 9, seed=1), picture(6, 9, seed=2)
    st = jpeg.stack_frames([a, b], "cpu")
    assert st.dtype == torch.uint8 and tuple(st.shape) == (2, 6, 9, 3)
    assert np.array_equal(st[0].numpy(), a) and np.array_equal(st[1].numpy(), b)
    assert np.array_equal(jpeg.stack_frames([a, b], "cpu", bgr=True)[1].numpy(), b[:, :, ::-1])
    rgba = np.concatenate([a, np.full((6, 9, 1), 255, np.uint8)], axis=2)
    assert np.array_equal(jpeg.to_device(rgba, "cpu").numpy(), a)
    t = torch.from_numpy(a.copy())
    assert jpeg.to_device(t, "cpu") is t
    assert np.array_equal(jpeg.to_device(t, "cpu", bgr=True).numpy(), a[:, :, ::-1])
    mixed = jpeg.stack_frames([a, torch.from_numpy(b.copy())], "cpu")
    assert np.array_equal(mixed[1].numpy(), b)
