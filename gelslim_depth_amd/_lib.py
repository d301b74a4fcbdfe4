"""ctypes binding of libgsd.so (include/gsd.h).  The product path: there is NO fallback.

If the shared library is missing or fails to load, importing this module raises; nothing in
gelslim_depth_amd/ ever routes to the oracle or to torch ops for the arithmetic of the hot path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSD_LIB_PATH") or os.path.join(_HERE, "csrc", "libgsd.so")   # override: A/B builds while tuning


class GsdError(RuntimeError):
    pass


class gsd_src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("off_h", C.c_int32), ("off_w", C.c_int32), ("relu", C.c_int32),
                ("w_stride", C.c_int32), ("slack", C.c_int32),
                ("n_stride", C.c_int64), ("c_stride", C.c_int64)]


class gsd_dst(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("off_h", C.c_int32), ("off_w", C.c_int32), ("w_stride", C.c_int32),
                ("n_stride", C.c_int64), ("c_stride", C.c_int64)]


class gsd_guard(C.Structure):
    """Non-finite guard of a train step (include/gsd.h): two device int32 words + the step's tick."""
    _fields_ = [("words", C.c_void_p), ("tick", C.c_int32)]


class gsd_nhwc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("pitch", C.c_int64),
                ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32)]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_D = C.c_double
_SRC = C.POINTER(gsd_src)
_DST = C.POINTER(gsd_dst)
_GUARD = C.POINTER(gsd_guard)
class gsd_bf16_bnbwd(C.Structure):
    _fields_ = [("y", C.POINTER(gsd_nhwc)), ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p),
                ("invstd", C.c_void_p)]


class gsd_wl_job(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wt", C.c_void_p), ("mode", C.c_int32), ("Co", C.c_int32), ("Ci", C.c_int32), ("reserved", C.c_int32)]


class gsd_bf16_wimg_job(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("mode", C.c_int32), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("reserved", C.c_int32)]


_NHWC = C.POINTER(gsd_nhwc)
_BNBWD = C.POINTER(gsd_bf16_bnbwd)
_IP = C.POINTER(C.c_int)

# name -> (restype, argtypes); mirrors include/gsd.h and include/gsd_bf16.h one to one (tests check every symbol loads)
SIGNATURES = {
    "gsd_version": (C.c_char_p, []),
    "gsd_last_error": (C.c_char_p, []),
    "gsd_selftest_mfma": (_I, [_P, _P, _P, _P]),
    "gsd_weight_layout_size": (_L, [_I, _I, _I]),
    "gsd_weight_layout": (_I, [_I, _P, _I, _I, _P, _P]),
    "gsd_weight_layout_batch": (_I, [C.POINTER(gsd_wl_job), _I, _P]),
    "gsd_conv3x3_partial_rows": (_I, [_I, _I, _I, _I]),
    "gsd_conv3x3": (_I, [_SRC, _I, _P, _I, _I, _DST, _I, _P, _I, _I, _I, _P]),
    "gsd_conv3x3_dgrad_bnrelu": (_I, [_SRC, _P, _I, _I, _DST, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "gsd_conv3x3_algo": (_I, [_I, _I, _I, _I, _I]),
    "gsd_conv3x3_w43_partial_rows": (_I, [_I, _I, _I, _I]),
    "gsd_conv3x3_w43_mfma_count": (C.c_int64, [_I, _I, _I, _I, _I]),
    "gsd_conv3x3_w43": (_I, [_SRC, _I, _P, _I, _I, _DST, _I, _P, _I, _I, _I, _P]),
    "gsd_conv3x3_w43_dgrad_bnrelu": (_I, [_SRC, _P, _I, _I, _DST, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "gsd_conv3x3_w2d_supported": (_I, [_I, _I]),
    "gsd_conv3x3_prefers_w2d": (_I, [_I, _I, _I, _I, _I, _I]),
    "gsd_conv3x3_w2d_estimate_us": (C.c_double, [_I, _I, _I, _I, _I]),
    "gsd_conv3x3_w43_estimate_us": (C.c_double, [_I, _I, _I, _I, _I, _I]),
    "gsd_conv3x3_w2d_partial_rows": (_I, [_I, _I, _I, _I]),
    "gsd_conv3x3_w2d_mfma_count": (C.c_int64, [_I, _I, _I, _I, _I]),
    "gsd_conv3x3_w2d": (_I, [_SRC, _I, _P, _I, _I, _DST, _I, _P, _I, _I, _I, _P]),
    "gsd_conv3x3_w2d_dgrad_bnrelu": (_I, [_SRC, _P, _I, _I, _DST, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "gsd_conv3x3_w2d_estimate_slabs_us": (C.c_double, [_I, _I, _I, _I, _I]),
    "gsd_conv3x3_w2d_workspace": (_L, [_I, _I, _I, _I, _I]),
    "gsd_conv3x3_w2d_ws": (_I, [_SRC, _I, _P, _I, _I, _DST, _I, _P, _P, _L, _I, _I, _I, _P]),
    "gsd_conv3x3_w2d_dgrad_bnrelu_ws": (_I, [_SRC, _P, _I, _I, _DST, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "gsd_conv3x3_w43_workspace": (_L, [_I, _I, _I, _I, _I]),
    "gsd_conv3x3_w43_ws": (_I, [_SRC, _I, _P, _I, _I, _DST, _I, _P, _P, _L, _I, _I, _I, _P]),
    "gsd_conv3x3_w43_dgrad_bnrelu_ws": (_I, [_SRC, _P, _I, _I, _DST, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    "gsd_convT2x2": (_I, [_SRC, _P, _P, _I, _I, _DST, _I, _I, _I, _P]),
    "gsd_convT2x2_dgrad_layout": (_I, [_SRC, _I, _I, _I, _I, _I]),
    "gsd_convT2x2_dgrad": (_I, [_SRC, _P, _I, _I, _DST, _I, _I, _I, _P]),
    "gsd_convT2x2_dgrad_as": (_I, [_I, _SRC, _P, _I, _I, _DST, _I, _I, _I, _P]),
    "gsd_convT2x2_dgrad_bnrelu_partial_rows": (_I, [_SRC, _I, _I, _I, _I, _I]),
    "gsd_convT2x2_dgrad_bnrelu": (_I, [_SRC, _P, _I, _I, _DST, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "gsd_conv3x3_wgrad_workspace": (_L, [_I, _I, _I, _I, _I]),
    "gsd_conv3x3_wgrad": (_I, [_SRC, _I, _SRC, _I, _I, _P, _P, _L, _I, _I, _I, _P]),
    "gsd_conv3x3_wgrad_form": (_I, [_SRC, _I, _SRC, _I, _I, _I, _I, _I]),
    "gsd_conv3x3_wgrad_mfma_count": (C.c_int64, [_I, _I, _I, _I, _I, _I]),
    "gsd_conv3x3_wgrad_takes_pitched_dy": (_I, [_I, _I, _I, _I, _I]),
    "gsd_conv3x3_wgrad_bn_supported": (_I, [_I, _I, _I, _I, _I]),
    "gsd_conv3x3_wgrad_bn_workspace": (_L, [_I, _I, _I, _I, _I]),
    "gsd_conv3x3_wgrad_bn": (_I, [_SRC, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _L, _I, _I, _I, _P]),
    "gsd_convT2x2_wgrad_workspace": (_L, [_I, _I, _I, _I, _I]),
    "gsd_convT2x2_wgrad": (_I, [_SRC, _SRC, _I, _I, _P, _P, _P, _L, _I, _I, _I, _P]),
    "gsd_bn_reduce_partials": (_I, [_P, _I, _I, _I, _P, _P]),
    "gsd_partials_channel_sums": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "gsd_add_counters": (_I, [C.POINTER(C.c_void_p), _I, _L, _P]),
    "gsd_bn_finalize": (_I, [_P, _I, _D, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _GUARD, _P]),
    "gsd_bn_eval_coeffs": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    "gsd_bn_bwd_partial_rows": (_I, [_I, _I, _I, _I]),
    "gsd_bn_bwd_reduce": (_I, [_I, _P, _P, _P, _P, _P, _SRC, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "gsd_bn_bwd_reduce_partials": (_I, [_P, _I, _I, _P, _P]),
    "gsd_bn_bwd_finalize": (_I, [_P, _P, _I, _D, _P, _P, _P, _P, _P, _P]),
    "gsd_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "gsd_sum_planes": (_I, [_P, _I, _I, _L, _P, _P, _P]),
    "gsd_maxpool2": (_I, [_SRC, _P, _I, _I, _I, _I, _P]),
    "gsd_conv1x1_out": (_I, [_SRC, _P, _P, _I, _I, _P, _I, _I, _I, _P]),
    "gsd_conv1x1_out_wgrad_rows": (_I, [_I, _I, _I]),
    "gsd_conv1x1_out_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P]),
    "gsd_loss_fwd_bwd": (_I, [_I, _P, _P, _L, _F, _P, _P, _P, _GUARD, _P]),
    "gsd_guard_snapshot": (_I, [_P, _P, _L, _P]),
    "gsd_guard_restore": (_I, [_GUARD, _P, _P, _L, _P]),
    "gsd_adam_ema": (_I, [_P, _P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _F, _F, _F, _GUARD, _P]),
    "gsd_bn_reduce_finalize": (_I, [_P, _I, _I, _I, _P, _D, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _GUARD, _P]),
    "gsd_bn_bwd_reduce_finalize": (_I, [_P, _I, _I, _I, _P, _D, _P, _P, _P, _P, _P, _P]),
    "gsd_area_resize_affine": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _I, _F, _F, _P]),
    "gsd_ingest_images": (_I, [_P, _P, _I, _I, _I, _I, _I, _L, _L, _L, _L, _P, _I, _I, _F, _F, _P]),
    "gsd_gaussian_blur": (_I, [_P, _L, _I, _I, _P, _I, _P, _P]),
    "gsd_channel_stats_workspace": (_L, [_I]),
    "gsd_channel_stats": (_I, [_P, _L, _I, _L, _P, _P, _P]),
    "gsd_gather_affine": (_I, [_P, _P, _L, _I, _I, _L, _P, _P, _I, _P, _P]),
    # ---- include/gsd_bf16.h
    "gsd_bf16_conv_mpad": (_I, [_I]),
    "gsd_bf16_conv_partial_rows": (_I, [_I, _I, _I, _I]),
    "gsd_bf16_conv_dense_partial_rows": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "gsd_bf16_conv3x3": (_I, [_NHWC, _P, _NHWC, _I, _I, _P, _BNBWD, _P]),
    "gsd_bf16_conv_dense": (_I, [_NHWC, _P, _NHWC, _I, _I, _I, _I, _IP, _IP, _I, _I, _I, _I, _I, _P, _P, _BNBWD, _P]),
    "gsd_bf16_conv3x3_bnrelu": (_I, [_NHWC, _P, _NHWC, _I, _I, _P, _P, _P]),
    "gsd_bf16_conv1x1_bnrelu": (_I, [_NHWC, _P, _NHWC, _I, _I, _P, _P, _P]),
    "gsd_bf16_weight_image_size": (_L, [_I, _I, _I]),
    "gsd_bf16_weight_image": (_I, [_I, _P, _I, _I, _P, _P]),
    "gsd_bf16_weight_images": (_I, [C.POINTER(gsd_bf16_wimg_job), _I, _P]),
    "gsd_bf16_im2col3x3": (_I, [_P, _I, _I, _I, _I, _NHWC, _P]),
    "gsd_bf16_conv3x3_first_supported": (_I, [_I, _I]),
    "gsd_bf16_conv3x3_first_partial_rows": (_I, [_I, _I, _I, _I]),
    "gsd_bf16_conv3x3_first": (_I, [_P, _I, _I, _I, _I, _P, _NHWC, _I, _P, _P, _P, _P]),
    "gsd_bf16_wgrad_first_workspace": (_L, [_I, _I, _I, _I]),
    "gsd_bf16_wgrad_first": (_I, [_P, _I, _I, _I, _I, _NHWC, _NHWC, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "gsd_bf16_inc_supported": (_I, [_I, _I]),
    "gsd_bf16_inc_conv_partial_rows": (_I, [_I, _I, _I]),
    "gsd_bf16_inc_conv": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _NHWC, _NHWC, _P, _P]),
    "gsd_bf16_first_bn_bwd_reduce": (_I, [_P, _I, _I, _I, _I, _P, _NHWC, _P, _P, _P, _P, _P, _P]),
    "gsd_bf16_wgrad_first_recompute": (_I, [_P, _I, _I, _I, _I, _P, _NHWC, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "gsd_bf16_conv3x3_c64_supported": (_I, [_I, _I]),
    "gsd_bf16_conv3x3_c64_partial_rows": (_I, [_I, _I, _I]),
    "gsd_bf16_conv3x3_c64": (_I, [_NHWC, _P, _NHWC, _P, _BNBWD, _P]),
    "gsd_bf16_bn_apply": (_I, [_NHWC, _P, _P, _NHWC, _I, _P]),
    "gsd_bf16_bn_apply_pool": (_I, [_NHWC, _P, _P, _NHWC, _NHWC, _P]),
    "gsd_bf16_bn_apply_pool_idx": (_I, [_NHWC, _P, _P, _NHWC, _NHWC, _P, _P]),
    "gsd_bf16_bn_bwd_reduce_pool_idx": (_I, [_NHWC, _P, _P, _P, _P, _NHWC, _P, _NHWC, _NHWC, _P, _P]),
    "gsd_bf16_maxpool2": (_I, [_NHWC, _NHWC, _P]),
    "gsd_bf16_conv1x1_out": (_I, [_NHWC, _P, _P, _I, _P, _P]),
    "gsd_bf16_bn_relu_conv1x1_out": (_I, [_NHWC, _P, _P, _P, _P, _I, _P, _P]),
    "gsd_bf16_bn_bwd_partial_rows": (_I, [_I, _I, _I]),
    "gsd_bf16_bn_bwd_reduce": (_I, [_I, _NHWC, _P, _P, _P, _P, _NHWC, _NHWC, _NHWC, _P, _P, _NHWC, _P, _P]),
    "gsd_bf16_bn_bwd_apply": (_I, [_NHWC, _NHWC, _P, _P, _P, _P, _P, _P]),
    "gsd_bf16_channel_sums_workspace": (_L, [_I, _I, _I, _I]),
    "gsd_bf16_channel_sums": (_I, [_NHWC, _I, _I, _I, _I, _P, _P, _L, _P]),
    "gsd_bf16_convT_bias_grad_workspace": (_L, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "gsd_bf16_convT_bias_grad": (_I, [_P, _I, _I, _I, _NHWC, _I, _I, _I, _I, _P, _P, _L, _P]),
    "gsd_bf16_wgrad_workspace": (_L, [_I, _I, _I, _I, _I, _I]),
    "gsd_bf16_wgrad": (_I, [_NHWC, _NHWC, _I, _I, _IP, _IP, _P, _I, _P, _L, _P]),
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise GsdError(
            f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or python -m gelslim_depth_amd.build). There is no CPU/torch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.gsd_last_error().decode("utf-8", "replace")
        raise GsdError(f"libgsd {what} failed (code {rc}): {msg}")


def version() -> str:
    return lib.gsd_version().decode()


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk_f32(t: torch.Tensor) -> None:
    if t.dtype != torch.float32 or not t.is_cuda:
        raise GsdError(f"expected a float32 CUDA(HIP) tensor, got {t.dtype} on {t.device}")


def _nchw_strides(t: torch.Tensor) -> Tuple[int, int, int]:
    """(n_stride, c_stride, w_stride) of an (N,C,H,W) tensor with unit column stride and non-overlapping rows / planes / images
    (contiguous, or a view of a pitched buffer).  torch leaves the stride of a size-1 dimension arbitrary (a permuted,
    channels_last or N == 1 tensor is still `is_contiguous()`): such a stride addresses nothing, so it is replaced by the
    dense one before the checks and before it is handed to a kernel."""
    if t.dim() != 4:
        raise GsdError(f"expected an NCHW tensor, got shape {tuple(t.shape)}")
    n, c, h, w = t.shape
    if w > 1 and t.stride(3) != 1:
        raise GsdError(f"expected an NCHW tensor with unit column stride, got shape {tuple(t.shape)} strides {t.stride()}")
    ws = t.stride(2) if (h > 1 or t.stride(2) >= w) else w              # a usable pitch is kept even when H == 1
    cs = t.stride(1) if (c > 1 or t.stride(1) >= h * ws) else h * ws
    ns = t.stride(0) if (n > 1 or t.stride(0) >= c * cs) else c * cs
    if ws < w or cs < h * ws or ns < c * cs:
        raise GsdError(f"expected an NCHW tensor with unit column stride, got shape {tuple(t.shape)} strides {t.stride()}")
    return ns, cs, ws


def _chk_nchw(t: torch.Tensor) -> None:
    _nchw_strides(t)


def pitched_empty(shape, device, pitch_multiple: int = 4) -> torch.Tensor:
    """An (N,C,H,W) fp32 tensor whose rows start 16-byte aligned: a view of a buffer with the row pitch rounded up to a
    multiple of 4 floats.  Kernels that take pitched operands move it as aligned 16-byte LDS-DMA pieces."""
    n, c, h, w = shape
    p = -(-w // pitch_multiple) * pitch_multiple
    return torch.empty((n, c, h, p), device=device, dtype=torch.float32)[..., :w]


def make_src(t: torch.Tensor, scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
             relu: bool = False, c_off: int = 0, c_len: Optional[int] = None,
             off: Tuple[int, int] = (0, 0), slack: int = 0) -> gsd_src:
    """Describe channels [c_off, c_off+c_len) of an NCHW tensor as a gsd_src segment.  The tensor may be a view of a
    PITCHED buffer (rows padded to a multiple of 4 floats): any strides with stride(3) == 1 are accepted.
    slack: readable floats the caller vouches for before and after the tensor (slack_empty allocates such tensors)."""
    _chk_f32(t)
    ns, cs, ws = _nchw_strides(t)
    n, ct, h, w = t.shape
    cl = ct - c_off if c_len is None else c_len
    s = gsd_src()
    s.ptr = t.data_ptr() + 4 * c_off * cs
    s.scale = ptr(scale)
    s.shift = ptr(shift)
    s.C, s.H, s.W = cl, h, w
    s.off_h, s.off_w = off
    s.relu = 1 if relu else 0
    s.w_stride = ws
    s.slack = slack
    s.n_stride = ns
    s.c_stride = cs
    return s


SLACK = 4   # floats of readable slack either side of a slack_empty tensor (gsd_src.slack)


def slack_empty(shape, device) -> torch.Tensor:
    """A contiguous fp32 tensor with SLACK readable floats in front of and behind it (16-byte aligned like any other): the
    Winograd dW kernel's 16-byte window pieces may read up to 3 floats past either end (gsd_src.slack)."""
    numel = 1
    for d in shape:
        numel *= int(d)
    buf = torch.empty((numel + 2 * SLACK,), device=device, dtype=torch.float32)
    buf[:SLACK].zero_()
    buf[-SLACK:].zero_()
    return buf[SLACK:SLACK + numel].view(*shape)


def make_dst(t: torch.Tensor, c_off: int = 0, c_len: Optional[int] = None, off: Tuple[int, int] = (0, 0)) -> gsd_dst:
    _chk_f32(t)
    ns, cs, ws = _nchw_strides(t)
    n, ct, h, w = t.shape
    cl = ct - c_off if c_len is None else c_len
    d = gsd_dst()
    d.ptr = t.data_ptr() + 4 * c_off * cs
    d.C, d.H, d.W = cl, h, w
    d.off_h, d.off_w = off
    d.w_stride = ws
    d.n_stride = ns
    d.c_stride = cs
    return d


def make_nhwc(t: torch.Tensor, c_off: int = 0, c_len: Optional[int] = None) -> gsd_nhwc:
    """Channels [c_off, c_off+c_len) of a contiguous (N,H,W,C) bfloat16 tensor as a gsd_nhwc view."""
    if t.dtype != torch.bfloat16 or not t.is_cuda or t.dim() != 4 or not t.is_contiguous():
        raise GsdError(f"expected a contiguous (N,H,W,C) bfloat16 CUDA(HIP) tensor, got {t.dtype} {tuple(t.shape)} on {t.device}")
    n, h, w, ct = t.shape
    d = gsd_nhwc()
    d.ptr = t.data_ptr() + 2 * c_off
    d.pitch = ct
    d.N, d.H, d.W = n, h, w
    d.C = ct - c_off if c_len is None else c_len
    return d


def make_guard(words: Optional[torch.Tensor], tick: int):
    """gsd_guard over a 2-element int32 device tensor, or None (a NULL guard)."""
    if words is None:
        return None
    if words.dtype != torch.int32 or not words.is_cuda or words.numel() < 2 or tick == 0:
        raise GsdError("a guard needs two int32 words on the GPU and a non-zero tick")
    g = gsd_guard()
    g.words = words.data_ptr()
    g.tick = tick
    return C.pointer(g)


def add_counters(tensors: Sequence[torch.Tensor], delta: int = 1) -> None:
    """counter += delta for every int64 scalar tensor on the GPU (BatchNorm2d.num_batches_tracked), one libgsd launch."""
    if not tensors:
        return
    for t in tensors:
        if t.dtype != torch.int64 or not t.is_cuda or t.numel() != 1:
            raise GsdError(f"add_counters: expected int64 scalars on the GPU, got {t.dtype} {tuple(t.shape)} on {t.device}")
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    check(lib.gsd_add_counters(arr, len(tensors), delta, stream_ptr()), "add_counters")


def int_array(vals: Sequence[int]):
    return (C.c_int * len(vals))(*vals)


def src_array(items: Sequence[gsd_src]):
    arr = (gsd_src * len(items))()
    for i, it in enumerate(items):
        arr[i] = it
    return arr


def dst_array(items: Sequence[gsd_dst]):
    arr = (gsd_dst * len(items))()
    for i, it in enumerate(items):
        arr[i] = it
    return arr
