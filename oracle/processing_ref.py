"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the inference pre/post-processing
around the model call: /root/reference/test_utils/test_depth_estimation.py:14-20 and the helpers it calls
(gelslim_depth/processing_utils/image_utils.py:6-15, normalization_utils.py:4-35,101-129).
Pinned by tests/test_oracle.py::test_processing against tests/golden/gproc.npz (made with the reference's own
normalisers and torch's F.interpolate(mode='area'))."""
import math

import numpy as np


def difference_image(a, b):
    return ((a - b + np.float32(255.0)) / np.float32(2.0)).astype(np.float32)       # image_utils.py:6-10


def area_resize(x, size):
    """F.interpolate(mode='area') == adaptive_avg_pool2d: window [floor(i*H/OH), ceil((i+1)*H/OH))."""
    n, c, h, w = x.shape
    oh, ow = size
    out = np.empty((n, c, oh, ow), np.float32)
    for i in range(oh):
        h0, h1 = (i * h) // oh, -((-(i + 1) * h) // oh)
        for j in range(ow):
            w0, w1 = (j * w) // ow, -((-(j + 1) * w) // ow)
            out[:, :, i, j] = x[:, :, h0:h1, w0:w1].mean(axis=(2, 3), dtype=np.float64)
    return out


def normalize_tactile(x, method, norm_scale, params=None):
    if "0_255" not in method:
        mins, maxes, means, stds = params
    if method == "min_max_to_-1_1":
        scale, bias, den = norm_scale, [0.5 * (a + b) for a, b in zip(maxes, mins)], [a - b for a, b in zip(maxes, mins)]
    elif method == "mean_std":
        scale, bias, den = 1.0, means, stds
    elif method == "0_255_to_-1_1":
        scale, bias, den = 2.0, [127.5], [255.0]
    elif method == "0_255_to_0_1":
        scale, bias, den = 1.0, [0.0], [255.0]
    out = np.zeros_like(x)
    for i in range(x.shape[1]):
        out[:, i] = scale * (x[:, i] - bias[min(i, len(bias) - 1)]) / den[min(i, len(den) - 1)]
    return out.astype(np.float32)


def denormalize_depth(d, method, norm_scale, params=None):
    vals = list(params) if params is not None else []
    mn, mx, mean, std = (vals + [None] * 4)[:4]
    if method == "min_max_to_-1_1":
        scale, bias, den = norm_scale, 0.5 * (mx + mn), mx - mn
    elif method == "mean_std":
        scale, bias, den = 1.0, mean, std
    elif method == "min_max_to_0_1":
        scale, bias, den = norm_scale, mn, mx - mn
    elif method == "min_max_to_0_-1":
        scale, bias, den = -norm_scale, mn, mx - mn
    return ((d * np.float32(den)) / np.float32(scale) + np.float32(bias)).astype(np.float32)
