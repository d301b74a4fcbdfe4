"""Build-owned deterministic generators for synthetic inputs and U-Net weights.

Nothing here comes from the reference: it exists so that golden vectors made in
the build container (tests/golden/make_golden.py, which imports the reference
model) can be compared on the GPU box against weights/inputs that are
*regenerated* there from a seed instead of shipped (124 MB for the full net).

numpy's PCG64 bit stream is stable across platforms for a fixed numpy version,
and the image pins numpy, so `numpy.random.Generator(PCG64(seed))` is the
generator of record.

Shapes/names follow the reference's state_dict layout
(/root/reference/gelslim_depth/models/unet.py:7-88).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Sequence

import numpy as np


def unet_state_shapes(n_channels: int, n_classes: int, layer_dimensions: Sequence[int],
                      kernel_size: int = 3) -> "OrderedDict[str, tuple]":
    """Ordered (name -> shape) map of the reference U-Net's state_dict.

    Order matches nn.Module registration order in the reference
    (unet.py:67 inc, :69-71 down[i], :73-75 up[i], :77 outc), which is also the
    order of `.parameters()` that Adam/EMA zip by position (train_unet.py:306,309).
    """
    dims = list(layer_dimensions)
    k = kernel_size
    out: "OrderedDict[str, tuple]" = OrderedDict()

    def double_conv(prefix: str, cin: int, cout: int, kk: int) -> None:
        out[f"{prefix}.double_conv.0.weight"] = (cout, cin, kk, kk)
        for bn in (1, 4):
            if bn == 4:
                out[f"{prefix}.double_conv.3.weight"] = (cout, cout, kk, kk)
            out[f"{prefix}.double_conv.{bn}.weight"] = (cout,)
            out[f"{prefix}.double_conv.{bn}.bias"] = (cout,)
            out[f"{prefix}.double_conv.{bn}.running_mean"] = (cout,)
            out[f"{prefix}.double_conv.{bn}.running_var"] = (cout,)
            out[f"{prefix}.double_conv.{bn}.num_batches_tracked"] = ()

    double_conv("inc", n_channels, dims[0], k)
    for i in range(len(dims) - 1):
        double_conv(f"down.{i}.maxpool_conv.1", dims[i], dims[i + 1], k)
    for j, i in enumerate(range(len(dims) - 1, 0, -1)):
        cin, cout = dims[i], dims[i - 1]
        out[f"up.{j}.up.weight"] = (cin, cin // 2, k - 1, k - 1)
        out[f"up.{j}.up.bias"] = (cin // 2,)
        double_conv(f"up.{j}.conv", cin, cout, 3)   # Up's DoubleConv always k=3 (unet.py:37)
    out["outc.conv.weight"] = (n_classes, dims[0], 1, 1)
    out["outc.conv.bias"] = (n_classes,)
    return out


def make_state(n_channels: int, n_classes: int, layer_dimensions: Sequence[int], seed: int,
               init: str = "conditioned") -> "OrderedDict[str, np.ndarray]":
    """Deterministic state_dict (numpy, fp32 / int64) for the U-Net.

    init="conditioned": He-scaled conv weights, BN gamma in [0.5,1.5], beta in
        [-0.2,0.2], running_mean ~ N(0,0.1), running_var in [0.5,1.5]; keeps
        activations O(1) through all levels so that a relative-L1 check on the
        output is meaningful (the reference's own init makes it vacuous).
    init="reference": every '*weight' ~ N(0, 0.01) (train_unet.py:248-250), BN
        beta 0, running stats (0,1), conv-transpose / outc biases U(+-1/sqrt(fan_in)).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = unet_state_shapes(n_channels, n_classes, layer_dimensions)
    st: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shp in shapes.items():
        if name.endswith("num_batches_tracked"):
            st[name] = np.zeros((), dtype=np.int64)
            continue
        is_bn = len(shp) == 1 and ".double_conv." in name
        if init == "reference":
            if name.endswith("weight"):
                v = rng.standard_normal(shp, dtype=np.float32) * np.float32(0.01)
            elif name.endswith("running_var"):
                v = np.ones(shp, np.float32)
            elif is_bn:
                v = np.zeros(shp, np.float32)
            else:  # up.*.up.bias / outc.conv.bias: U(+-1/sqrt(fan_in))
                wshape = shapes[name[:-4] + "weight"]
                if name.startswith("up."):   # ConvTranspose2d fan_in = weight.size(1)*k*k
                    fan_in = wshape[1] * wshape[2] * wshape[3]
                else:
                    fan_in = wshape[1] * wshape[2] * wshape[3]
                b = 1.0 / np.sqrt(fan_in)
                v = rng.uniform(-b, b, shp).astype(np.float32)
        else:
            if len(shp) == 4:
                if name.startswith("up.") and name.endswith("up.weight"):
                    fan_in = shp[0]            # each output pixel sums Cin products
                else:
                    fan_in = shp[1] * shp[2] * shp[3]
                std = np.sqrt(2.0 / fan_in)
                if name.startswith("outc"):
                    std = np.sqrt(1.0 / fan_in)
                v = rng.standard_normal(shp, dtype=np.float32) * np.float32(std)
            elif name.endswith("running_mean"):
                v = rng.standard_normal(shp, dtype=np.float32) * np.float32(0.1)
            elif name.endswith("running_var"):
                v = rng.uniform(0.5, 1.5, shp).astype(np.float32)
            elif is_bn and name.endswith("weight"):
                v = rng.uniform(0.5, 1.5, shp).astype(np.float32)
            elif is_bn:
                v = rng.uniform(-0.2, 0.2, shp).astype(np.float32)
            else:
                v = rng.uniform(-0.1, 0.1, shp).astype(np.float32)
        st[name] = np.ascontiguousarray(v, dtype=np.float32)
    return st


def make_batch(n: int, h: int, w: int, seed: int, n_channels: int = 3, n_classes: int = 1):
    """Synthetic (x, target): x ~ U[0,1) like a /255 difference image
    (image_utils.py:9, normalization_utils.py:19-22); target ~ U(-0.9,0], the range of
    'min_max_to_0_-1' * norm_scale 0.9 (normalization_utils.py:93-98, train_unet.py:38)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.random((n, n_channels, h, w), dtype=np.float32)
    t = -np.float32(0.9) * rng.random((n, n_classes, h, w), dtype=np.float32)
    return x, t


def param_names(state_names: List[str]) -> List[str]:
    """Names that are nn.Parameters (everything except BN running stats / counters)."""
    return [n for n in state_names
            if not (n.endswith("running_mean") or n.endswith("running_var")
                    or n.endswith("num_batches_tracked"))]
