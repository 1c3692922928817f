"""ctypes binding of libpremvos_hip.so -- the only way the Python host reaches the GPU kernels.

There is NO fallback: if the library is missing and cannot be built (hipcc absent) every op
raises.  torch is imported first so the library binds to the HIP runtime torch already loaded
(same SONAME), which makes torch's stream handles valid inside the library.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL: shares libamdhip64 with the library)

from . import build as _build

_LIB = None
ABI_VERSION = 18         # premvos_abi_version() of the library this file's SIGNATURES / ConvDesc describe

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3
ACT_SPLIT8_BF16 = 0x200      # premvos_dwconv3x3_f32: store the resident S8 layout ({hi8, lo8} per group of 8 channels) for premvos_conv_bf16x3_s8_f32
OUT_NHWC, OUT_PIXSHUF2 = 0, 1
PREC_F32, PREC_BF16, PREC_BF16X3 = 0, 1, 3
PRECISIONS = {"fp32": PREC_F32, "bf16": PREC_BF16, "bf16x3": PREC_BF16X3}


class ConvDesc(C.Structure):
    """Mirror of ``premvos_conv_desc`` (include/premvos_hip.h)."""
    _fields_ = [
        ("inp", C.c_void_p), ("wgt", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
        ("out", C.c_void_p),
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32),
        ("in_ps", C.c_int32),
        ("ho", C.c_int32), ("wo", C.c_int32), ("cout", C.c_int32),
        ("out_ps", C.c_int32), ("res_ps", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
        ("dh", C.c_int32), ("dw", C.c_int32), ("pt", C.c_int32), ("pl", C.c_int32),
        ("cin_pad", C.c_int32), ("k_pad", C.c_int32), ("cout_pad", C.c_int32),
        ("act", C.c_int32), ("slope", C.c_float), ("out_mode", C.c_int32), ("cout_ps", C.c_int32),
        ("tile_hint", C.c_int32), ("split_k", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("precision", C.c_int32), ("stage_k", C.c_int32), ("wgt_lo", C.c_void_p),
        ("tail_m_tiles", C.c_int32), ("tail_split_k", C.c_int32), ("wgt_wino", C.c_void_p),
        ("wgt_wino4", C.c_void_p),
    ]


_i32, _f32, _vp = C.c_int32, C.c_float, C.c_void_p

# name -> argtypes; every symbol include/premvos_hip.h declares (tests check the header against this)
SIGNATURES = {
    "premvos_conv2d_f32": [C.POINTER(ConvDesc), _vp],
    "premvos_corr_fwd_f32": [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "premvos_corr_nchw_fwd_f32": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "premvos_warp_corr_fwd_f32": [_vp, _i32, _vp, _i32, _vp, _i32, _f32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32,
                                  _i32, _vp],
    "premvos_warp_fwd_f32": [_vp, _i32, _vp, _i32, _f32, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "premvos_nchw_to_nhwc_f32": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "premvos_nhwc_to_nchw_f32": [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp],
    "premvos_flow_preprocess_u8": [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp],
    "premvos_flow_postprocess_f32": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp],
    "premvos_proposal_preprocess_u8": [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp],
    "premvos_maxpool_f32": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "premvos_rpn_proposals_f32": [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _f32, _f32, _f32, _i32, _i32,
                                  _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp],
    "premvos_roi_align_f32": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _f32, _i32, _vp, _i32, _vp],
    "premvos_global_avgpool_f32": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "premvos_refine_input_u8": [_vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp],
    "premvos_dwconv3x3_f32": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32,
                              _i32, _i32, _i32, _i32, _vp],
    "premvos_conv_bf16x3_s8_f32": [C.POINTER(ConvDesc), _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp],
    "premvos_split8_f32": [_vp, _i32, _vp, _i32, C.c_int64, _i32, _vp],
    "premvos_resize_bilinear_f32": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp],
    "premvos_broadcast_pixel_f32": [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp],
    "premvos_refine_output_f32": [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "premvos_mfma_f32_calibrate": [C.c_int64, _i32, _vp, _vp],
    "premvos_mfma_f32_calibrate_random": [C.c_int64, _i32, _vp, _vp],
    "premvos_hbm_copy_calibrate": [_vp, _vp, C.c_int64, _vp],
    "premvos_digest_u64": [_vp, C.c_int64, _i32, _i32, _vp, _vp],
    "premvos_conv_wino4_slab_f32": [C.POINTER(ConvDesc), _vp, C.c_int64, _i32, _i32, _i32, _vp],
    "premvos_jpeg_entropy_decode_host": [_vp, C.c_int64, _vp, _vp, C.c_int64],
    "premvos_jpeg_reconstruct_u8": [_vp, _vp, _vp, _vp, _i32, _vp],
    "premvos_reid_input_u8": [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp],
    "premvos_scale_shift_relu_f32": [_vp, _i32, C.c_int64, _i32, _vp, _vp, _vp, _i32, _i32, _vp],
    "premvos_mask_warp_u8": [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp],
    "premvos_mask_overlap_u8": [_vp, _i32, _vp, _i32, C.c_int64, _vp, _vp, _vp, _vp],
    "premvos_mask_pack_bits_u8": [_vp, C.c_int64, _vp, _vp],
    "premvos_mask_unpack_bits_u8": [_vp, C.c_int64, _vp, _vp],
    "premvos_rle_boundaries_u8": [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp],
    "premvos_rle_boundaries_pooled_u8": [_vp, _i32, _i32, _i32, C.c_int64, _i32, _vp, _i32, _vp, _vp, _vp],
    "premvos_frcnn_tail_f32": [_vp, _i32, _vp, _vp, _i32, _i32, _f32, _f32, _f32, _f32, _i32, _f32, _f32, _f32, _f32,
                               _f32, _vp, _vp, _vp, _vp, _vp],
}


class FrameFiles(C.Structure):
    """premvos_frame_files (include/premvos_hip.h): the arguments of premvos_write_frame_files_host."""
    _fields_ = [("flo_path", C.c_char_p), ("flow", C.c_void_p), ("flow_row_stride", C.c_int64), ("h", C.c_int32), ("w", C.c_int32),
                ("boxes", C.c_void_p * 2), ("probs", C.c_void_p * 2), ("count", C.c_int32 * 2), ("scale", C.c_float),
                ("json_path", C.c_char_p * 4), ("conf", C.c_void_p), ("rle_pool", C.c_void_p), ("rle_offsets", C.c_void_p)]


class PremvosError(RuntimeError):
    pass


def lib_path() -> str:
    return _build.LIB


def load():
    """Load (building first if sources are newer and hipcc exists).  Raises if impossible."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    stale = None
    try:
        if _build.needs_build():
            # one builder at a time (every rank of a torchrun job lands here): an exclusive file lock around the check + build
            import fcntl
            with open(os.path.join(os.path.dirname(path), ".build.lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                if _build.needs_build():
                    _build.build_lib()
    except Exception as e:  # no hipcc on this box: a prebuilt .so must have travelled
        if not os.path.exists(path):
            raise PremvosError(
                f"libpremvos_hip.so is missing and could not be built ({e}); the HIP path has no fallback") from e
        stale = e
    lib = C.CDLL(path)
    lib.premvos_last_error.restype = C.c_char_p
    lib.premvos_last_error.argtypes = []
    lib.premvos_abi_version.restype = C.c_int
    lib.premvos_abi_version.argtypes = []
    if lib.premvos_abi_version() != ABI_VERSION:
        # a stale binary with another struct layout / argument list would corrupt arguments silently
        raise PremvosError(f"{path} has ABI version {lib.premvos_abi_version()}, this package binds version {ABI_VERSION}"
                           + (f" (rebuilding failed: {stale})" if stale else "") + "; rebuild with python -m premvos_amd.build")
    if stale is not None:
        import warnings
        warnings.warn(f"libpremvos_hip.so is older than its sources and could not be rebuilt ({stale}); using it as is")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.premvos_conv2d_workspace_bytes.argtypes = [C.POINTER(ConvDesc)]
    lib.premvos_conv2d_workspace_bytes.restype = C.c_int64
    lib.premvos_crc32c_host.argtypes = [_vp, C.c_int64]
    lib.premvos_crc32c_host.restype = C.c_uint32
    lib.premvos_refine_output_workspace_bytes.argtypes = [_i32, _i32, _i32, _i32]
    lib.premvos_refine_output_workspace_bytes.restype = C.c_int64
    lib.premvos_rle_counts_to_string_host.argtypes = [_vp, C.c_int64, _vp, C.c_int64]
    lib.premvos_rle_counts_to_string_host.restype = C.c_int64
    lib.premvos_rle_strings_host.argtypes = [_vp, _vp, _i32, C.c_int64, _vp, C.c_int64, _vp]
    lib.premvos_rle_strings_host.restype = C.c_int64
    lib.premvos_write_frame_files_host.argtypes = [C.POINTER(FrameFiles)]
    lib.premvos_write_frame_files_host.restype = C.c_int
    lib.premvos_format_floats_host.argtypes = [_vp, C.c_int64, _i32, _vp, C.c_int64]
    lib.premvos_format_floats_host.restype = C.c_int
    lib.premvos_rle_workspace_bytes.argtypes = [_i32, _i32, _i32]
    lib.premvos_rle_workspace_bytes.restype = C.c_int64
    lib.premvos_jpeg_workspace_bytes.argtypes = [_vp]
    lib.premvos_jpeg_workspace_bytes.restype = C.c_int64
    _LIB = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        raise PremvosError(f"{what} failed ({rc}): {load().premvos_last_error().decode()}")


def current_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def resolve_device(device=None) -> "torch.device":
    """``device`` with its INDEX: None / "cuda" / an index-less torch.device mean "the calling thread's current GPU" -- which is a
    per-thread setting (a new thread starts on device 0), so an object that keeps such a value and is later used from another thread
    would allocate on another GPU.  Every constructor / helper of this package resolves the device ONCE, here, on the thread that
    calls it, and passes the indexed device on (tests/test_cpu_device_discipline.py: no bare "cuda" elsewhere in premvos_amd/).
    A tensor stands for its own device.  Without a visible GPU the value passes through (host-only tests)."""
    if isinstance(device, torch.Tensor):
        device = device.device
    d = torch.device("cuda" if device is None else device)
    if d.type == "cuda" and d.index is None and torch.cuda.is_available():
        d = torch.device("cuda", torch.cuda.current_device())
    return d


def require_gpu():
    if not torch.cuda.is_available():
        raise PremvosError("no GPU visible: premvos_amd runs only on the HIP path (no CPU fallback)")
