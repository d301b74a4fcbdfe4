"""Drop-in for the reference's `gelslim_depth.models.unet.UNet`, computed by libgsd (MI355X HIP kernels).

Boundary reproduced (reference file:line, /root/reference/):
  * constructor  UNet(n_channels, n_classes, layer_dimensions=[64,128,256,512,1024], kernel_size=3,
                 maxpool_size=2, upconv_stride=2, bilinear=False)          gelslim_depth/models/unet.py:61
  * attributes   n_channels, n_classes, bilinear                           unet.py:63-65
  * forward(x)   accepts the `x=` keyword (train_utils/train_unet.py:347,
                 test_utils/test_depth_estimation.py:17); returns fp32 (N, n_classes, H, W) on x's device,
                 differentiable w.r.t. the parameters in train mode
  * state_dict   the reference's 118 keys / shapes / dtypes, so its .pth files load with strict=True
                 (test_depth_estimation.py:63): the module tree below has the same attribute names and
                 Sequential indices (double_conv.{0,1,3,4}, maxpool_conv.1, up.{j}.up, up.{j}.conv, outc.conv)
  * parameter order == registration order (Adam / EMA zip by position, train_unet.py:306,309), names of all
    weights contain 'weight' (init loop train_unet.py:248-250)

The leaves are parameter holders only: the arithmetic of the whole network is ONE autograd node whose
forward/backward are libgsd kernel schedules (engine.UNetEngine).  Calling a leaf's forward raises --
there is deliberately no torch-op fallback.  Only the configuration the reference's own scripts use is
implemented by the kernels: kernel_size=3 (padding is hard-wired to 1 at unet.py:11,14), maxpool_size=2,
upconv_stride=2 (=> ConvTranspose2d k=2, s=2).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn as nn

from ..engine import UNetEngine


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError(f"{type(self).__name__} only holds parameters; call UNet.forward (libgsd computes the "
                           "whole network, there is no per-layer torch fallback)")


class _ConvParams(_Holder):
    """Parameters of nn.Conv2d / nn.ConvTranspose2d with torch's default initialisation."""

    def __init__(self, wshape: Sequence[int], bias_len: int, fan_in: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*wshape))
        # torch: kaiming_uniform_(a=sqrt(5)) over fan_in = weight.size(1) * receptive field (also for ConvTranspose2d)
        bound_w = math.sqrt(6.0 / ((1 + 5.0) * fan_in))
        with torch.no_grad():
            self.weight.uniform_(-bound_w, bound_w)
        if bias_len:
            self.bias = nn.Parameter(torch.empty(bias_len))
            b = 1.0 / math.sqrt(fan_in)
            with torch.no_grad():
                self.bias.uniform_(-b, b)


class _BatchNormParams(_Holder):
    """Parameters and buffers of nn.BatchNorm2d (eps 1e-5, momentum 0.1, affine, track_running_stats)."""

    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Slot(_Holder):
    """Parameter-less position in a Sequential (ReLU at indices 2/5, MaxPool2d at index 0)."""


class DoubleConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3):
        super().__init__()
        k = kernel_size
        self.double_conv = nn.Sequential(
            _ConvParams((out_channels, in_channels, k, k), 0, in_channels * k * k),
            _BatchNormParams(out_channels),
            _Slot(),
            _ConvParams((out_channels, out_channels, k, k), 0, out_channels * k * k),
            _BatchNormParams(out_channels),
            _Slot(),
        )


class Down(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, maxpool_size: int = 2):
        super().__init__()
        self.maxpool_conv = nn.Sequential(_Slot(), DoubleConv(in_channels, out_channels, kernel_size=kernel_size))


class Up(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 2, stride: int = 2):
        super().__init__()
        k = kernel_size
        # ConvTranspose2d weight is (Cin, Cout, k, k); torch computes its fan_in from weight.size(1)
        self.up = _ConvParams((in_channels, in_channels // 2, k, k), in_channels // 2, (in_channels // 2) * k * k)
        self.conv = DoubleConv(in_channels, out_channels)


class OutConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = _ConvParams((out_channels, in_channels, 1, 1), out_channels, in_channels)


class _UNetFunction(torch.autograd.Function):
    """The whole U-Net as one autograd node: forward = engine.forward, backward = engine.backward."""

    @staticmethod
    def forward(ctx, module: "UNet", x: torch.Tensor, *params: torch.Tensor):
        P = module._tensor_map()
        out = module._engine.forward(x, P, train=True)
        ctx.module = module
        ctx.pnames = module._pnames
        ctx.engine = module._engine
        ctx.generation = module._engine.generation
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        module: "UNet" = ctx.module
        # The activations live in the engine's buffers, not in ctx: a later forward (another batch, an eval pass, gradient
        # accumulation over two forwards) has overwritten them, and back-propagating through them would be silently wrong.
        if module._engine is not ctx.engine or module._engine.generation != ctx.generation:
            raise RuntimeError(
                "gelslim_depth_amd.UNet: backward() of a forward whose saved activations are gone -- the model ran another "
                f"forward since (forward #{ctx.generation}, engine is at #{module._engine.generation}) or changed precision. "
                "Call backward() before the next forward(); for gradient accumulation, accumulate .grad between "
                "forward/backward pairs.")
        P = module._tensor_map()
        G = module._grad_targets()
        module._engine.backward(dout, P, G)
        return (None, None) + tuple(G[n] for n in ctx.pnames)


class UNet(nn.Module):
    def __init__(self, n_channels, n_classes, layer_dimensions=[64, 128, 256, 512, 1024], kernel_size=3,
                 maxpool_size=2, upconv_stride=2, bilinear=False, precision="fp32"):
        super().__init__()
        if kernel_size != 3 or maxpool_size != 2 or upconv_stride != 2:
            raise NotImplementedError(
                "libgsd implements the configuration the reference trains and ships (kernel_size=3, maxpool_size=2, "
                f"upconv_stride=2); got kernel_size={kernel_size}, maxpool_size={maxpool_size}, "
                f"upconv_stride={upconv_stride}")
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.bilinear = bilinear
        dims = list(layer_dimensions)

        self.inc = DoubleConv(n_channels, dims[0], kernel_size=kernel_size)
        self.down = nn.ModuleList()
        for i in range(len(dims) - 1):
            self.down.append(Down(dims[i], dims[i + 1], kernel_size=kernel_size, maxpool_size=maxpool_size))
        self.up = nn.ModuleList()
        for i in range(len(dims) - 1, 0, -1):
            self.up.append(Up(dims[i], dims[i - 1], kernel_size=kernel_size - 1, stride=upconv_stride))
        self.outc = OutConv(dims[0], n_classes)

        self._dims = dims
        self._engine = None
        self.set_precision(precision)
        self._pnames: List[str] = [n for n, _ in self.named_parameters()]
        # optional flat gradient arena installed by TrainStep: name -> view
        self._grad_views: Dict[str, torch.Tensor] = {}

    def set_precision(self, precision: str) -> "UNet":
        """"fp32" (default; the reference's arithmetic, BASELINE configs[1-3]) or "bf16" (mixed precision, configs[4]:
        bf16 NHWC activations/gradients on the bf16 MFMA, fp32 master weights / statistics / optimiser state)."""
        if precision == "fp32":
            eng = UNetEngine(self.n_channels, self.n_classes, self._dims)
        elif precision == "bf16":
            from ..engine_bf16 import UNetEngineBF16
            eng = UNetEngineBF16(self.n_channels, self.n_classes, self._dims)
        else:
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
        if self._engine is not None:            # keep the data-parallel hooks a TrainStep installed
            eng.world, eng.sync_fn = self._engine.world, self._engine.sync_fn
        self._engine = eng
        self.precision = precision
        return self

    # -- plumbing -------------------------------------------------------------------------------
    def _tensor_map(self) -> Dict[str, torch.Tensor]:
        m = {n: p.data for n, p in self.named_parameters()}
        m.update({n: b for n, b in self.named_buffers()})
        return m

    def _grad_targets(self) -> Dict[str, torch.Tensor]:
        if self._grad_views:
            return self._grad_views
        return {n: torch.empty_like(p.data) for n, p in self.named_parameters()}

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("gelslim_depth_amd.UNet runs on an MI355X through libgsd; move the model and the input "
                               "to the GPU (there is no CPU path in this package)")
        if x.requires_grad and torch.is_grad_enabled():
            # the reference module is differentiable w.r.t. its input (unet.py:79-88); nothing in the reference asks for that
            # gradient (train_unet.py:344-347) and the first convolution's dX is not built: say so instead of returning None
            raise NotImplementedError("gelslim_depth_amd.UNet does not compute the gradient with respect to its input "
                                      "(x.requires_grad is set); detach x, or use the reference model for input gradients")
        x = x.float()
        if self.training and torch.is_grad_enabled():
            return _UNetFunction.apply(self, x, *self.parameters())
        with torch.no_grad():
            return self._engine.forward(x, self._tensor_map(), train=self.training)
