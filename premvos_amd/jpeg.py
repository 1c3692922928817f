"""Optional GPU JPEG decode (``PREMVOS_GPU_JPEG=1``): the frame a stage reads arrives in HBM as the bytes
its reference reader would have produced -- ``cv2.imread`` (proposal_net/train.py:500, BGR), ``scipy.ndimage.imread`` = PIL
(optical_flow_net-PWC-Net/script_pwc_multi.py:34) or PIL (ReID_net/prepare_input.py:38) -- without the CPU running the inverse
DCT, the chroma up-sampling and the colour conversion, and with ONE upload per frame for all stages.

Two halves, matching the C-ABI (include/premvos_hip.h, csrc/jpeg_ops.hip):

  ``host_stage(fn_or_bytes)``   marker parsing + Huffman decoding on a host thread (plain C++, the GIL is released during
                                the call; the bit stream is serial) into pinned quantised coefficients;
  ``device_stage(decoded)``     upload + ``premvos_jpeg_reconstruct_u8`` on the current HIP stream -> uint8 [H,W,3] in HBM.

Files the decoder does not cover (progressive, arithmetic-coded, 12-bit, CMYK / RGB-coded, multi-scan, 4:4:0 / 4:1:1) keep the
default reader: ``host_stage`` returns the PIL-decoded array for them and ``device_stage`` uploads it.  The default of every
driver is unchanged (PIL on the host); this is the optional fast path SURVEY asks to keep optional.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Dict, List, Tuple, Union

import numpy as np
import torch

from . import _lib

EUNSUPPORTED = -3


class JpegInfo(C.Structure):
    """premvos_jpeg_info of include/premvos_hip.h."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("ncomp", C.c_int32), ("hs", C.c_int32), ("vs", C.c_int32),
                ("mcux", C.c_int32), ("mcuy", C.c_int32), ("blocks_w", C.c_int32 * 3), ("blocks_h", C.c_int32 * 3),
                ("reserved", C.c_int32), ("coef_offset", C.c_int64 * 3), ("coef_count", C.c_int64),
                ("quant", (C.c_uint16 * 64) * 3)]


class Unsupported(ValueError):
    """A valid JPEG outside what the GPU decoder covers."""


def enabled(default: str = "0") -> bool:
    """PREMVOS_GPU_JPEG=1 / 0; unset: ``default`` (the stage drivers keep the library reader of the reference's scripts, the
    streaming driver turns the GPU path on -- same bytes either way, tests/test_gpu_jpeg.py)."""
    return os.environ.get("PREMVOS_GPU_JPEG", default) not in ("", "0")


class _PinnedPool:
    """Pinned int16 staging buffers, reused once the upload that read them has finished."""

    def __init__(self):
        self.lock = threading.Lock()
        self.free: Dict[int, List[Tuple[torch.Tensor, object]]] = {}

    def get(self, count: int) -> torch.Tensor:
        size = max(1 << 16, 1 << (int(count) - 1).bit_length())
        with self.lock:
            lst = self.free.get(size, [])
            for i, (t, ev) in enumerate(lst):
                if ev is None or ev.query():
                    lst.pop(i)
                    return t
        pin = torch.cuda.is_available()
        return torch.empty(size, dtype=torch.int16, pin_memory=pin)

    def put(self, t: torch.Tensor, ev) -> None:
        with self.lock:
            lst = self.free.setdefault(t.numel(), [])
            if len(lst) < 64:
                lst.append((t, ev))


_POOL = _PinnedPool()


class Decoded:
    """The host half's result: geometry / tables and the quantised coefficients in a pinned buffer."""
    __slots__ = ("info", "coef")

    def __init__(self, info: JpegInfo, coef: torch.Tensor):
        self.info, self.coef = info, coef

    @property
    def shape(self) -> Tuple[int, int, int]:
        return (self.info.height, self.info.width, 3)


def _read(src: Union[str, bytes, bytearray, memoryview]) -> bytes:
    if isinstance(src, (bytes, bytearray, memoryview)):
        return bytes(src)
    with open(src, "rb") as f:
        return f.read()


def header(data: bytes) -> JpegInfo:
    info = JpegInfo()
    rc = _lib.load().premvos_jpeg_entropy_decode_host(data, len(data), C.byref(info), None, 0)
    if rc == EUNSUPPORTED:
        raise Unsupported(_lib.load().premvos_last_error().decode())
    _lib.check(rc, "premvos_jpeg_entropy_decode_host")
    return info


def entropy_decode(data: bytes) -> Decoded:
    """Huffman-decode a baseline JPEG on the host.  Raises ``Unsupported`` for files the decoder does not cover."""
    lib = _lib.load()
    info = header(data)
    coef = _POOL.get(info.coef_count)
    rc = lib.premvos_jpeg_entropy_decode_host(data, len(data), C.byref(info), coef.data_ptr(), coef.numel())
    if rc != 0:
        _POOL.put(coef, None)
        _lib.check(rc, "premvos_jpeg_entropy_decode_host")
    return Decoded(info, coef)


def reconstruct(d: Decoded, device=None, bgr: bool = False) -> torch.Tensor:
    """Upload the coefficients and run the two kernels on the current stream of ``device``: uint8 [H,W,3]."""
    device = _lib.resolve_device(device)
    _lib.require_gpu()
    lib, info = _lib.load(), d.info
    if d.coef is None:
        raise ValueError("this frame's coefficients were already handed to the GPU (reconstruct() consumes a Decoded)")
    n = int(info.coef_count)
    coef_dev = torch.empty(n, dtype=torch.int16, device=device)
    coef_dev.copy_(d.coef[:n], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _POOL.put(d.coef, ev)
    d.coef = None
    ws = torch.empty(max(8, lib.premvos_jpeg_workspace_bytes(C.byref(info))), dtype=torch.uint8, device=device)
    out = torch.empty((info.height, info.width, 3), dtype=torch.uint8, device=device)
    _lib.check(lib.premvos_jpeg_reconstruct_u8(coef_dev.data_ptr(), C.byref(info), ws.data_ptr(), out.data_ptr(), int(bool(bgr)),
                                               _lib.current_stream()), "premvos_jpeg_reconstruct_u8")
    return out


def decode(data: bytes, device=None, bgr: bool = False) -> torch.Tensor:
    return reconstruct(entropy_decode(data), device, bgr)


# ---- the two halves as the drivers use them -------------------------------------------------------------------------------
def _pil_rgb(src) -> np.ndarray:
    import io
    from PIL import Image
    im = Image.open(io.BytesIO(src) if isinstance(src, (bytes, bytearray)) else src)
    return np.array(im.convert("RGB"))[:, :, :3]        # (a writable copy: torch.from_numpy wants one)


def host_stage(src: Union[str, bytes]) -> Union[Decoded, np.ndarray]:
    """What a decode-ahead thread does for one frame: the entropy decode (GPU path) or, for a file the GPU decoder does not cover
    (and for anything that is not a JPEG), the default reader's RGB array."""
    data = _read(src)
    if data[:2] != b"\xff\xd8":
        return _pil_rgb(data)
    try:
        return entropy_decode(data)
    except Unsupported:
        return _pil_rgb(data)


def device_stage(item: Union[Decoded, np.ndarray], device=None, bgr: bool = False) -> torch.Tensor:
    """uint8 [H,W,3] in HBM on the current stream, RGB (or BGR, the cv2.imread order)."""
    if isinstance(item, Decoded):
        return reconstruct(item, device, bgr)
    t = torch.from_numpy(item).to(_lib.resolve_device(device))
    return t.flip(2).contiguous() if bgr else t


def imread(src: Union[str, bytes], device=None, bgr: bool = False) -> torch.Tensor:
    return device_stage(host_stage(src), device, bgr)


def to_device(item, device=None, bgr: bool = False) -> torch.Tensor:
    """One frame as the drivers hold it -- a host RGB array, a ``Decoded`` (entropy-decoded, not yet reconstructed) or a uint8 tensor
    already in HBM -- as uint8 [H,W,3] on ``device`` (current stream)."""
    if isinstance(item, torch.Tensor):
        t = item if item.is_cuda else item.to(_lib.resolve_device(device))
        if item.is_cuda:
            # decoded on another stream (and complete: its producer synchronised); tell the allocator this stream reads it too,
            # so the block is not handed out again while a kernel of this stream is still in flight
            item.record_stream(torch.cuda.current_stream(item.device))
        return t.flip(2).contiguous() if bgr else t
    if isinstance(item, Decoded):
        return reconstruct(item, device, bgr)
    return device_stage(np.array(item[:, :, :3], dtype=np.uint8, order="C"), device, bgr)


def stack_frames(items, device=None, bgr: bool = False) -> torch.Tensor:
    """Frames of equal size -> uint8 [n,H,W,3] on ``device``: ONE upload for host arrays, the GPU decode for ``Decoded`` items."""
    items = list(items)
    if all(isinstance(i, np.ndarray) for i in items):
        t = torch.from_numpy(np.stack([np.array(i[:, :, :3], dtype=np.uint8, order="C") for i in items])).to(_lib.resolve_device(device))
        return t.flip(3).contiguous() if bgr else t
    return torch.stack([to_device(i, device, bgr) for i in items])


def loader(default: str = "0"):
    """The per-frame function of a driver's decode-ahead pool: file name -> host RGB array (default) or, with PREMVOS_GPU_JPEG=1,
    the entropy-decoded frame for ``stack_frames`` / ``to_device`` to finish on the GPU."""
    return host_stage if enabled(default) else _pil_rgb
