"""Inference pre/post-processing around the model call, on the GPU through libgsd.

Mirrors (paths under /root/reference/):
  get_difference_image                       gelslim_depth/processing_utils/image_utils.py:6-10
  sample_multi_channel_image_to_desired_size image_utils.py:12-15   (F.interpolate(mode='area'))
  normalize_tactile_image                    processing_utils/normalization_utils.py:4-35
  denormalize_depth_image                    normalization_utils.py:101-129
  predict_depth_from_RGB                     test_utils/test_depth_estimation.py:14-20  (the working copy; the library
                                             copy complete_prediction.py:4-10 reads config attributes no config defines)

resize -> normalise -> model(x=...) -> de-normalise -> resize runs as: ONE kernel (difference image + area resize +
per-channel affine), the libgsd U-Net, ONE kernel (per-channel affine + area resize).  Averaging and an affine map
commute, so fusing them changes only the rounding order.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L
from ._lib import lib, check


def tactile_affine(method: str, norm_scale: float, params=None) -> Tuple[Sequence[float], Sequence[float]]:
    """(A, B) per channel with normalize_tactile_image(x) == A*x + B  (normalization_utils.py:4-35)."""
    if "0_255" not in method:
        mins, maxes, means, stds = params
    if method == "min_max_to_-1_1":
        scale, bias, den = norm_scale, [0.5 * (a + b) for a, b in zip(maxes, mins)], [a - b for a, b in zip(maxes, mins)]
    elif method == "mean_std":
        scale, bias, den = 1.0, list(means), list(stds)
    elif method == "0_255_to_-1_1":
        scale, bias, den = 2.0, [127.5], [255.0]
    elif method == "0_255_to_0_1":
        scale, bias, den = 1.0, [0.0], [255.0]
    else:
        raise ValueError(f"unknown image_normalization_method {method!r}")
    n = max(len(bias), len(den))
    A = [scale / den[min(i, len(den) - 1)] for i in range(n)]
    B = [-scale * bias[min(i, len(bias) - 1)] / den[min(i, len(den) - 1)] for i in range(n)]
    return A, B


def depth_denorm_affine(method: str, norm_scale: float, params=None) -> Tuple[float, float]:
    """(A, B) with denormalize_depth_image(d) == A*d + B  (normalization_utils.py:101-129)."""
    mn = mx = mean = std = None
    if "0_255" not in method:
        vals = list(params)
        mn = vals[0] if len(vals) > 0 else None
        mx = vals[1] if len(vals) > 1 else None
        mean = vals[2] if len(vals) > 2 else None
        std = vals[3] if len(vals) > 3 else None
    if method == "min_max_to_-1_1":
        scale, bias, den = norm_scale, 0.5 * (mx + mn), mx - mn
    elif method == "mean_std":
        scale, bias, den = 1.0, mean, std
    elif method == "min_max_to_0_1":
        scale, bias, den = norm_scale, mn, mx - mn
    elif method == "min_max_to_0_-1":
        scale, bias, den = -norm_scale, mn, mx - mn
    else:
        raise ValueError(f"unknown depth_normalization_method {method!r}")
    return den / scale, bias


def area_resize_affine(x: torch.Tensor, size: Tuple[int, int], A: Sequence[float], B: Sequence[float],
                       base: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = A[c] * area_resize(pre(x)) + B[c];  pre(x) = (x - base + 255)/2 when base is given."""
    if not x.is_cuda:
        raise L.GsdError("area_resize_affine needs GPU tensors (no CPU path in this package)")
    x = x.float().contiguous()
    n, c, h, w = x.shape
    oh, ow = int(size[0]), int(size[1])
    a = torch.tensor(list(A), device=x.device, dtype=torch.float32)
    b = torch.tensor(list(B), device=x.device, dtype=torch.float32)
    bs = None
    if base is not None:
        bs = base.float().expand_as(x).contiguous()
    out = torch.empty((n, c, oh, ow), device=x.device, dtype=torch.float32)
    check(lib.gsd_area_resize_affine(x.data_ptr(), L.ptr(bs), n, c, h, w, out.data_ptr(), oh, ow, a.data_ptr(), b.data_ptr(),
                                     a.numel(), 255.0, 0.5, L.stream_ptr()), "area_resize_affine")
    return out


def get_difference_image(tactile_image: torch.Tensor, base_tactile_image: torch.Tensor) -> torch.Tensor:
    """(tactile - base + 255) / 2   (image_utils.py:6-10)."""
    return area_resize_affine(tactile_image, tactile_image.shape[-2:], [1.0], [0.0], base=base_tactile_image)


def sample_multi_channel_image_to_desired_size(MC_image: torch.Tensor, desired_size: Tuple[int, int],
                                               interp_method: str = "area") -> torch.Tensor:
    """F.interpolate(MC_image, size=desired_size, mode='area')  (image_utils.py:12-15)."""
    if interp_method != "area":
        raise NotImplementedError("only interp_method='area' (what the reference's configs use) is implemented")
    return area_resize_affine(MC_image, desired_size, [1.0], [0.0])


def predict_depth_from_RGB(images: torch.Tensor, model, output_size: Tuple[int, int], config,
                           base_images: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Drop-in for test_depth_estimation.py:14-20.  `images` are (difference) images in 0..255; with `base_images`
    the difference image (test_depth_estimation.py:83) is folded into the first kernel."""
    A, B = tactile_affine(config.image_normalization_method, config.norm_scale,
                          getattr(config, "image_normalization_parameters", None))
    x = area_resize_affine(images, config.input_tactile_image_size, A, B, base=base_images)
    depth = model(x=x)
    dA, dB = depth_denorm_affine(config.depth_normalization_method, config.norm_scale,
                                 config.depth_normalization_parameters)
    return area_resize_affine(depth, output_size, [dA], [dB])
