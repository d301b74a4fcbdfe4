"""Host-side schedule of the U-Net hot path in bf16 mixed precision (BASELINE.json configs[4]) over include/gsd_bf16.h.

Same interface as engine.UNetEngine (forward / backward / sync_fn / block_done_cb / kernel_log), same reference control
flow (/root/reference/gelslim_depth/models/unet.py:79-88), different data layout:

  * activations and their gradients: bfloat16, NHWC.  Per conv unit u:
        y[u]  raw convolution output           a[u]  relu(bn(y)) -- materialised (2 B/element, one extra HBM pass that
        g[u]  gradient: da -> dz -> dy in place       costs ~4 % of the step and lets every GEMM operand go HBM -> LDS by DMA)
  * torch.cat([skip, up]) (unet.py:48) is ONE buffer cat[l] of C_skip + C_up channels per level: the encoder's BatchNorm
    apply writes a[skip] into channels [0, C_skip), the transposed convolution scatters its output (+bias) into channels
    [C_skip, ..) at its F.pad offset (unet.py:43-47; the padding border is zeroed once), the decoder conv reads the
    buffer as one tensor.  gcat[l] is its gradient: one dX launch, whose two channel slices feed the skip unit's
    BatchNorm backward and the transposed convolution's dX / dW.
  * fp32: master parameters (the module's own tensors), every accumulation, BatchNorm statistics, gradients, Adam/EMA.
    Per step each weight is re-laid-out to its bf16 GEMM image (gsd_bf16_weight_image).
  * eval mode: BatchNorm (running statistics) + ReLU ride in the conv epilogue (gsd_bf16_conv3x3_bnrelu): y is not stored.
  * backward: pass 1 of BatchNorm+ReLU backward (mask, per-channel sums) is fused into the dX launch that produces the
    gradient (conv3x3 and transposed-conv dX); only skip units (gradient = dX slice + max-pool routing) and the last unit
    (gradient = OutConv backward) use the stand-alone reduce kernels.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from ._lib import lib, check

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
T3Y = [t // 3 - 1 for t in range(9)]
T3X = [t % 3 - 1 for t in range(9)]


def _r32(c: int) -> int:
    return (c + 31) // 32 * 32


class _Unit:
    """conv3x3(no bias) + BatchNorm2d + ReLU (unet.py:11-13 / :14-16)."""

    def __init__(self, prefix: str, conv_idx: int, bn_idx: int, cin: int, cout: int, level: int):
        self.cin, self.cout, self.level = cin, cout, level
        self.wname = f"{prefix}.double_conv.{conv_idx}.weight"
        bn = f"{prefix}.double_conv.{bn_idx}."
        self.gname, self.bname = bn + "weight", bn + "bias"
        self.rmname, self.rvname, self.nbtname = bn + "running_mean", bn + "running_var", bn + "num_batches_tracked"
        self.first = False          # the im2col'd first layer
        self.need_dgrad = True
        self.wt_f = self.wt_d = None
        self.scale = self.shift = self.mean = self.invstd = self.c1 = self.c2 = self.sums = None
        self.fused_rows = 0         # partial rows written by the dX launch whose epilogue did pass 1 of this unit's BatchNorm backward
        self.y = self.g = None      # (N,H,W,Cout) bf16
        self.a = None               # gsd_nhwc view of the activation (own tensor or a slice of a concat buffer)
        self.a_t = None             # tensor backing `a`
        self.a_off = 0
        self.src = None             # (tensor, c_off, c_len) of the unit's input, kept for the weight gradient


class _Up:
    """ConvTranspose2d(cin, cin//2, 2, 2) (unet.py:36)."""

    def __init__(self, j: int, cin: int, level_in: int):
        self.j, self.cin, self.cout, self.level_in = j, cin, cin // 2, level_in
        self.wname, self.bname = f"up.{j}.up.weight", f"up.{j}.up.bias"
        self.wt_f = self.wt_d = None


class UNetEngineBF16:
    precision = "bf16"

    def __init__(self, n_channels: int, n_classes: int, layer_dimensions: Sequence[int]):
        dims = list(layer_dimensions)
        if any(d % 32 for d in dims):
            raise ValueError(f"bf16 path needs layer_dimensions that are multiples of 32 (MFMA k-step), got {dims}")
        if n_classes != 1:
            raise NotImplementedError("bf16 path implements n_classes == 1 (what every reference config uses)")
        self.n_channels, self.n_classes, self.dims = n_channels, n_classes, dims
        self.L = len(dims) - 1
        self.enc: List[Tuple[_Unit, _Unit]] = []
        self.dec: List[Tuple[_Unit, _Unit]] = []
        self.ups: List[_Up] = []
        self.enc.append((_Unit("inc", 0, 1, n_channels, dims[0], 0), _Unit("inc", 3, 4, dims[0], dims[0], 0)))
        self.enc[0][0].first = True
        self.enc[0][0].need_dgrad = False
        for i in range(self.L):
            p = f"down.{i}.maxpool_conv.1"
            self.enc.append((_Unit(p, 0, 1, dims[i], dims[i + 1], i + 1), _Unit(p, 3, 4, dims[i + 1], dims[i + 1], i + 1)))
        for j, i in enumerate(range(self.L, 0, -1)):
            cin, cout = dims[i], dims[i - 1]
            if dims[i - 1] + cin // 2 != cin:
                raise ValueError(f"layer_dimensions {dims}: level {i} needs dims[i-1] + dims[i]//2 == dims[i] "
                                 "(the reference model fails at torch.cat/conv otherwise)")
            self.ups.append(_Up(j, cin, i))
            p = f"up.{j}.conv"
            self.dec.append((_Unit(p, 0, 1, cin, cout, i - 1), _Unit(p, 3, 4, cout, cout, i - 1)))
        self.units: List[_Unit] = [u for pair in self.enc for u in pair] + [u for pair in self.dec for u in pair]
        self._shape = None
        self.sync_fn: Optional[Callable[[torch.Tensor], None]] = None
        self.world = 1
        self._saved_train = False
        self.block_done_cb: Optional[Callable[[str], None]] = None
        self.guard = None          # non-finite guard of the current step (_lib.make_guard), set by TrainStep per step
        self.generation = 0        # forwards so far: a backward belongs to exactly one (models/unet.py checks it)
        self.kernel_log: Optional[list] = None
        self.region_log: Optional[list] = None     # bench hook: (region name, start event, end event)

    # ------------------------------------------------------------------ buffers
    # tuning switches the library re-reads on every call and that change how many partial rows a launch writes: the buffers below
    # are sized from them, so they are part of the shape key -- a switch flipped between two steps (a test's monkeypatch, a sweep
    # in one process) re-sizes the buffers instead of overrunning them
    _SIZING_ENV = ("GSD_BF16_BN_BLOCKS", "GSD_BF16_CTGEMM", "GSD_BF16_CT_BM", "GSD_BF16_TW", "GSD_BF16_XCD")

    def _ensure(self, n: int, h: int, w: int, dev: torch.device, train: bool) -> None:
        import os
        key = (n, h, w, str(dev), tuple(os.environ.get(k) for k in self._SIZING_ENV), train)
        if self._shape == key:
            return
        if self._shape is not None and self._shape[:5] == key[:5] and not train:
            return                      # eval after train at the same shape: everything needed exists
        self._shape = key
        hs, ws = [h], [w]
        for _ in range(self.L):
            hs.append(hs[-1] // 2)
            ws.append(ws[-1] // 2)
        assert hs[-1] >= 1 and ws[-1] >= 1, "input too small for this many max-pools"
        self.hs, self.ws = hs, ws
        bf = dict(device=dev, dtype=torch.bfloat16)
        f32 = dict(device=dev, dtype=torch.float32)
        # first layer: straight from x (gsd_bf16_conv3x3_first / gsd_bf16_wgrad_first) where the shape is served, else through the
        # im2col'd input (col0) and the dense-tap kernels; GSD_BF16_FIRST=0 forces the im2col path
        self._wready, self._wevents, self._wdone = {}, {}, set()
        self.first_direct = bool(lib.gsd_bf16_conv3x3_first_supported(self.n_channels, self.dims[0])) and \
            os.environ.get("GSD_BF16_FIRST", "1") != "0"
        self.col0 = None if self.first_direct else torch.empty((n, h, w, _r32(9 * self.n_channels)), **bf)
        # train mode: the `inc` double convolution without its first convolution's raw output in HBM (gsd_bf16_inc.hip: statistics
        # from a write-free pass over x, the 64 -> 64 kernel rebuilds relu(bn(conv(x))) per halo tile, the backward recomputes the
        # raw output from x); GSD_BF16_FUSED_INC=0: the four unfused launches
        self.fused_inc = self.first_direct and bool(lib.gsd_bf16_inc_supported(self.n_channels, self.dims[0])) and \
            os.environ.get("GSD_BF16_FUSED_INC", "1") != "0"
        # concat buffers (zeroed once: the F.pad border of the `up` slice is never written again)
        self.cat = [torch.zeros((n, hs[l], ws[l], self.dims[l] + self.dims[l + 1] // 2), **bf) for l in range(self.L)]
        self.gcat = [torch.empty_like(c) for c in self.cat] if train else [None] * self.L
        self.pooled = [None] + [torch.empty((n, hs[l], ws[l], self.dims[l - 1]), **bf) for l in range(1, self.L + 1)]
        self.dpooled = [None] + ([torch.empty_like(self.pooled[l]) for l in range(1, self.L + 1)] if train else [None] * self.L)
        max_part, max_ws = 1, 64
        for u in self.units:
            self._alloc_unit(u, n, dev, train)
        skips = {id(self.enc[l][1]) for l in range(self.L)}
        for l in range(self.L):         # skip connection: the activation lives in the concat buffer
            self.enc[l][1].a_t, self.enc[l][1].a_off = self.cat[l], 0
        for u in self.units:
            if id(u) not in skips and (u.a_t is None or tuple(u.a_t.shape) != (n, hs[u.level], ws[u.level], u.cout)):
                u.a_t, u.a_off = torch.empty((n, hs[u.level], ws[u.level], u.cout), **bf), 0
        for u in self.units:
            lh, lw = hs[u.level], ws[u.level]
            u.a = L.make_nhwc(u.a_t, u.a_off, u.cout)
            mp = lib.gsd_bf16_conv_mpad(u.cout)
            direct = u.first and self.first_direct
            rows = lib.gsd_bf16_conv3x3_first_partial_rows(n, lh, lw, u.cout) if direct else lib.gsd_bf16_conv_partial_rows(n, lh, lw, u.cout)
            max_part = max(max_part, rows * 2 * mp)
            if self.fused_inc and u is self.enc[0][1]:
                max_part = max(max_part, lib.gsd_bf16_inc_conv_partial_rows(n, lh, lw) * 2 * mp)
            if train:
                max_part = max(max_part, lib.gsd_bf16_bn_bwd_partial_rows(n, lh, lw) * 3 * u.cout)
                if direct:
                    max_ws = max(max_ws, lib.gsd_bf16_wgrad_first_workspace(n, lh, lw, u.cout))
                else:
                    kcols = _r32(9 * u.cin) if u.first else u.cin
                    max_ws = max(max_ws, lib.gsd_bf16_wgrad_workspace(1 if u.first else 9, n, lh, lw, u.cout, kcols))
        for up in self.ups:
            li = up.level_in
            if up.wt_f is None or up.wt_f.device != dev:
                up.wt_f = torch.empty((lib.gsd_bf16_weight_image_size(3, up.cout, up.cin),), **bf)
                up.wt_d = torch.empty((lib.gsd_bf16_weight_image_size(4, up.cout, up.cin),), **bf)
            if train:
                # the transposed convolution's dX carries pass 1 of the BatchNorm backward of the unit below it
                max_part = max(max_part, lib.gsd_bf16_conv_dense_partial_rows(n, hs[li], ws[li], up.cout, up.cin, 4, 2) * 2 *
                               lib.gsd_bf16_conv_mpad(up.cin))
                max_ws = max(max_ws, lib.gsd_bf16_wgrad_workspace(4, n, hs[li], ws[li], up.cin, up.cout))
                max_ws = max(max_ws, lib.gsd_bf16_channel_sums_workspace(n, 2 * hs[li], 2 * ws[li], up.cout))
                oy_, ox_ = self._pad_off(li - 1)
                max_ws = max(max_ws, lib.gsd_bf16_convT_bias_grad_workspace(n, hs[li - 1], ws[li - 1], oy_, ox_, 2 * hs[li], 2 * ws[li], up.cout))
        self.partials = torch.empty((max_part,), **f32)
        self.wspace = torch.empty((max(max_ws, 64),), **f32) if train else None
        # weight gradients on a SIDE stream: dW(u) only needs d_raw(u) and the unit's input, nothing downstream of it waits for
        # it, and it is MFMA-bound while the backward chain it leaves behind alternates with HBM-bound BatchNorm passes (20 % of
        # the bf16 step): the two overlap where neither fills the chip.  Own split-K workspace; joined before a block's
        # gradients are handed to the all-reduce and at the end of backward.  GSD_BF16_SIDE_DW=0: everything on one stream.
        # BatchNorm apply + max-pool of the encoder's skip units in one pass (gsd_bf16_bn_apply_pool); GSD_BF16_APPLY_POOL=0: two
        self.apply_pool = os.environ.get("GSD_BF16_APPLY_POOL", "1") != "0"
        # ... and that pass leaves the pool's arg-max (2 bits per element) for the backward, which then does not re-read the
        # window's activations (gsd_bf16_bn_apply_pool_idx / gsd_bf16_bn_bwd_reduce_pool_idx); GSD_BF16_POOL_IDX=0: it does
        self.pool_index = train and self.apply_pool and os.environ.get("GSD_BF16_POOL_IDX", "1") != "0"
        self.pool_idx = [None] + [torch.empty((n, hs[l], ws[l], self.dims[l - 1] // 8), device=dev, dtype=torch.int16)
                                  if self.pool_index else None for l in range(1, self.L + 1)]
        # 64 -> 64 convolutions (forward and dX) on the weights-resident kernel (gsd_bf16_c64.hip); GSD_BF16_C64=0: the DMA-filled one
        self.c64 = os.environ.get("GSD_BF16_C64", "1") != "0"
        # train mode: the last unit's BatchNorm + ReLU rides in the 1x1 output convolution (gsd_bf16_bn_relu_conv1x1_out): its
        # activation has no other reader (the backward recomputes it from the raw output) and is never written.  GSD_BF16_FUSED_OUT=0: apply + conv
        self.fused_out = os.environ.get("GSD_BF16_FUSED_OUT", "1") != "0"
        # the transposed convolutions' bias gradient from the statistics rows of the dX launch that writes the gradient slice (the
        # decoder's first convolution) instead of a pass over the slice (gsd_bf16_convT_bias_grad); GSD_BF16_DB_FROM_DX=0: gsd_bf16_channel_sums
        self.db_from_dx = train and os.environ.get("GSD_BF16_DB_FROM_DX", "1") != "0"
        self.db_part = [torch.empty((lib.gsd_bf16_conv_partial_rows(n, hs[l], ws[l], self.cat[l].shape[3]) * 2 *
                                     lib.gsd_bf16_conv_mpad(self.cat[l].shape[3]),), **f32) if self.db_from_dx else None
                        for l in range(self.L)]
        self.side_dw = train and os.environ.get("GSD_BF16_SIDE_DW", "1") != "0"
        self.side = torch.cuda.Stream(device=dev) if self.side_dw else None
        self.wspace_side = torch.empty((max(max_ws, 64),), **f32) if self.side_dw else None

    def _alloc_unit(self, u: _Unit, n: int, dev: torch.device, train: bool) -> None:
        lh, lw = self.hs[u.level], self.ws[u.level]
        bf = dict(device=dev, dtype=torch.bfloat16)
        f32 = dict(device=dev, dtype=torch.float32)
        if u.y is None or tuple(u.y.shape) != (n, lh, lw, u.cout):
            u.y = torch.empty((n, lh, lw, u.cout), **bf)
            u.g = None
            u.a_t = None
        if train and u.g is None:
            u.g = torch.empty((n, lh, lw, u.cout), **bf)
        if u.scale is None or u.scale.device != dev:
            for nm in ("scale", "shift", "mean", "invstd", "c1", "c2"):
                setattr(u, nm, torch.empty((u.cout,), **f32))
            u.sums = torch.empty((65 * 3 * u.cout,), device=dev, dtype=torch.float64)
            u.wt_f = torch.empty((lib.gsd_bf16_weight_image_size(2 if u.first else 0, u.cout, u.cin),), **bf)
            u.wt_d = torch.empty((lib.gsd_bf16_weight_image_size(1, u.cout, u.cin),), **bf) if u.need_dgrad else None

    # ------------------------------------------------------------------ helpers
    def _pad_off(self, lvl: int) -> Tuple[int, int]:
        # F.pad(x1, [dX//2, dX-dX//2, dY//2, dY-dY//2]) (unet.py:43-47)
        dy = self.hs[lvl] - 2 * self.hs[lvl + 1]
        dx = self.ws[lvl] - 2 * self.ws[lvl + 1]
        return dy // 2, dx // 2

    def _region_begin(self):
        if self.region_log is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _region_end(self, name: str, e0) -> None:
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.region_log.append((name, e0, e1))

    def _log(self, name: str, flops: float, sig=None):
        """bench hook: returns a closer that records (name, flops, start, end, shape signature)."""
        if self.kernel_log is None:
            return lambda: None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()

        def close():
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.kernel_log.append((name, flops, e0, e1, sig))
        return close

    def _run_unit(self, u: _Unit, src: Tuple[torch.Tensor, int, int], P, train: bool, st: int,
                  pool_to: Optional[torch.Tensor] = None) -> None:
        """pool_to (train mode): the unit's activation feeds a max-pool -- BatchNorm apply and the pool are one pass over y."""
        n, lh, lw = u.y.shape[0], self.hs[u.level], self.ws[u.level]
        u.src = src
        din = L.make_nhwc(*src) if src[0] is not None else None     # the first layer's direct kernels read x itself
        dy = L.make_nhwc(u.y)
        part = self.partials.data_ptr()
        if not train:
            # eval: BatchNorm uses the running statistics, known before the convolution -> conv + BN + ReLU in ONE kernel,
            # straight into the activation (no raw output, no separate apply pass)
            check(lib.gsd_bn_eval_coeffs(P[u.gname].data_ptr(), P[u.bname].data_ptr(), P[u.rmname].data_ptr(),
                                         P[u.rvname].data_ptr(), BN_EPS, u.cout, u.scale.data_ptr(), u.shift.data_ptr(), st),
                  "bn_eval_coeffs")
            if u.first and self.first_direct:
                self._wimage(2, P[u.wname], u.cout, u.cin, u.wt_f, st)
                check(lib.gsd_bf16_conv3x3_first(self._x.data_ptr(), n, u.cin, lh, lw, u.wt_f.data_ptr(), C.byref(u.a), u.cout, None,
                                                 u.scale.data_ptr(), u.shift.data_ptr(), st), "conv3x3_first")
            elif u.first:
                self._wimage(2, P[u.wname], u.cout, u.cin, u.wt_f, st)
                check(lib.gsd_bf16_conv1x1_bnrelu(C.byref(din), u.wt_f.data_ptr(), C.byref(u.a), src[2], u.cout, u.scale.data_ptr(),
                                                  u.shift.data_ptr(), st), "conv1x1_bnrelu")
            else:
                self._wimage(0, P[u.wname], u.cout, u.cin, u.wt_f, st)
                done = self._log("bf16_conv3x3", 2.0 * u.cout * u.cin * 9 * n * lh * lw, (u.cout, u.cin, lh, lw))
                check(lib.gsd_bf16_conv3x3_bnrelu(C.byref(din), u.wt_f.data_ptr(), C.byref(u.a), u.cin, u.cout, u.scale.data_ptr(),
                                                  u.shift.data_ptr(), st), "conv3x3_bnrelu")
                done()
            return
        if u.first and self.first_direct:
            self._wimage(2, P[u.wname], u.cout, u.cin, u.wt_f, st)
            done = self._log("bf16_conv_first", 2.0 * u.cout * 9 * u.cin * n * lh * lw)
            check(lib.gsd_bf16_conv3x3_first(self._x.data_ptr(), n, u.cin, lh, lw, u.wt_f.data_ptr(), C.byref(dy), u.cout, part, None,
                                             None, st), "conv3x3_first")
            done()
        elif u.first:
            self._wimage(2, P[u.wname], u.cout, u.cin, u.wt_f, st)
            z = L.int_array([0])
            done = self._log("bf16_conv_dense", 2.0 * u.cout * src[2] * n * lh * lw)
            check(lib.gsd_bf16_conv_dense(C.byref(din), u.wt_f.data_ptr(), C.byref(dy), src[2], u.cout, 1, 1, z, z, lh, lw, 0, 0, 0,
                                          None, part, None, st), "conv_dense(first)")
            done()
        elif self._use_c64(u.cin, u.cout):
            self._wimage(0, P[u.wname], u.cout, u.cin, u.wt_f, st)
            done = self._log("bf16_conv3x3", 2.0 * u.cout * u.cin * 9 * n * lh * lw, (u.cout, u.cin, lh, lw))
            check(lib.gsd_bf16_conv3x3_c64(C.byref(din), u.wt_f.data_ptr(), C.byref(dy), part, None, st), "conv3x3_c64")
            done()
            self._finalize_stats(u, lib.gsd_bf16_conv3x3_c64_partial_rows(n, lh, lw), float(n * lh * lw), P, st)
            self._apply(u, dy, st, pool_to)
            return
        else:
            self._wimage(0, P[u.wname], u.cout, u.cin, u.wt_f, st)
            done = self._log("bf16_conv3x3", 2.0 * u.cout * u.cin * 9 * n * lh * lw, (u.cout, u.cin, lh, lw))
            check(lib.gsd_bf16_conv3x3(C.byref(din), u.wt_f.data_ptr(), C.byref(dy), u.cin, u.cout, part, None, st), "conv3x3")
            done()
        rows = (lib.gsd_bf16_conv3x3_first_partial_rows(n, lh, lw, u.cout) if (u.first and self.first_direct)
                else lib.gsd_bf16_conv_partial_rows(n, lh, lw, u.cout))
        self._finalize_stats(u, rows, float(n * lh * lw), P, st)
        self._apply(u, dy, st, pool_to)

    def _use_c64(self, k: int, m: int) -> bool:
        # (the weights-resident kernel fills through buffer descriptors: one image of an operand must stay below 2 GiB)
        return self.c64 and bool(lib.gsd_bf16_conv3x3_c64_supported(k, m)) and self.hs[0] * self.ws[0] * 64 * 2 < (1 << 31)

    def _run_inc_fused(self, u0: _Unit, u1: _Unit, P, st: int, pool_to: Optional[torch.Tensor]) -> None:
        """Train-mode `inc` (unet.py:7-20, :67) without u0's raw output: statistics of conv(x) from a write-free pass, then ONE
        kernel that rebuilds relu(bn(conv(x))) per halo tile, writes it once (u1's dW reads it) and runs the 64 -> 64
        convolution on it from LDS (gsd_bf16_inc.hip)."""
        n, lh, lw = u1.y.shape[0], self.hs[0], self.ws[0]
        x = self._x
        u0.src, u1.src = (None, 0, 0), (u0.a_t, u0.a_off, u0.cout)
        part = self.partials.data_ptr()
        self._wimage(2, P[u0.wname], u0.cout, u0.cin, u0.wt_f, st)
        self._wimage(0, P[u1.wname], u1.cout, u1.cin, u1.wt_f, st)
        done = self._log("bf16_conv_first", 2.0 * u0.cout * 9 * u0.cin * n * lh * lw)
        check(lib.gsd_bf16_conv3x3_first(x.data_ptr(), n, u0.cin, lh, lw, u0.wt_f.data_ptr(), None, u0.cout, part, None, None, st),
              "conv3x3_first (statistics)")
        done()
        self._finalize_stats(u0, lib.gsd_bf16_conv3x3_first_partial_rows(n, lh, lw, u0.cout), float(n * lh * lw), P, st)
        dy1 = L.make_nhwc(u1.y)
        done = self._log("bf16_inc_fused", 2.0 * u1.cout * (u1.cin + u0.cin) * 9 * n * lh * lw)
        check(lib.gsd_bf16_inc_conv(x.data_ptr(), n, u0.cin, lh, lw, u0.wt_f.data_ptr(), u0.scale.data_ptr(), u0.shift.data_ptr(),
                                    u1.wt_f.data_ptr(), C.byref(u0.a), C.byref(dy1), part, st), "inc_conv")
        done()
        self._finalize_stats(u1, lib.gsd_bf16_inc_conv_partial_rows(n, lh, lw), float(n * lh * lw), P, st)
        self._apply(u1, dy1, st, pool_to)

    def _finalize_stats(self, u: _Unit, rows: int, count: float, P, st: int) -> None:
        """self.partials holds `rows` BatchNorm partial rows of unit u's raw output: batch statistics -> (mean, invstd, scale,
        shift), running statistics, the batch counter."""
        if self.sync_fn is None:     # a few hundred partial rows: column sums and finalize in ONE launch
            check(lib.gsd_bn_reduce_finalize(self.partials.data_ptr(), rows, lib.gsd_bf16_conv_mpad(u.cout), u.cout, u.sums.data_ptr(),
                                             count, P[u.gname].data_ptr(), P[u.bname].data_ptr(), BN_EPS, BN_MOMENTUM,
                                             P[u.rmname].data_ptr(), P[u.rvname].data_ptr(), u.mean.data_ptr(), u.invstd.data_ptr(),
                                             u.scale.data_ptr(), u.shift.data_ptr(), self.guard, st), "bn_reduce_finalize")
        else:                        # SyncBN: the fp64 sums are all-reduced between the two halves
            check(lib.gsd_bn_reduce_partials(self.partials.data_ptr(), rows, lib.gsd_bf16_conv_mpad(u.cout), u.cout,
                                             u.sums.data_ptr(), st), "bn_reduce_partials")
            self.sync_fn(u.sums[:2 * u.cout])
            count *= self.world
            check(lib.gsd_bn_finalize(u.sums.data_ptr(), u.cout, count, P[u.gname].data_ptr(), P[u.bname].data_ptr(),
                                      BN_EPS, BN_MOMENTUM, P[u.rmname].data_ptr(), P[u.rvname].data_ptr(),
                                      u.mean.data_ptr(), u.invstd.data_ptr(), u.scale.data_ptr(), u.shift.data_ptr(),
                                      self.guard, st),
                  "bn_finalize")
        self._nbt.append(P[u.nbtname])   # int64 counters: one libgsd launch for all of them at the end of the forward

    def _last_unit(self) -> _Unit:
        return self.dec[-1][1] if self.L > 0 else self.enc[0][1]

    def _apply(self, u: _Unit, dy, st: int, pool_to: Optional[torch.Tensor]) -> None:
        """train mode: a = relu(bn(y)) (+ the max-pool of a skip unit in the same pass)."""
        if self.fused_out and u is self._last_unit():
            return      # forward() folds it into the output convolution
        if pool_to is not None:
            dp = L.make_nhwc(pool_to)
            idx = self.pool_idx[u.level + 1] if self.pool_index else None
            check(lib.gsd_bf16_bn_apply_pool_idx(C.byref(dy), u.scale.data_ptr(), u.shift.data_ptr(), C.byref(u.a), C.byref(dp),
                                                 L.ptr(idx), st), "bn_apply_pool")
        else:
            check(lib.gsd_bf16_bn_apply(C.byref(dy), u.scale.data_ptr(), u.shift.data_ptr(), C.byref(u.a), 1, st), "bn_apply")

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, P: Dict[str, torch.Tensor], train: bool, out: Optional[torch.Tensor] = None
                ) -> torch.Tensor:
        if x.dtype != torch.float32 or not x.is_cuda:
            raise L.GsdError("UNetEngineBF16.forward needs a float32 tensor on the GPU (no CPU fallback)")
        x = x.contiguous()
        n, c, h, w = x.shape
        assert c == self.n_channels, f"expected {self.n_channels} input channels, got {c}"
        self._ensure(n, h, w, x.device, train)
        st = L.stream_ptr()
        self._saved_train = train
        self._nbt = []
        self.generation += 1       # every forward overwrites the saved activations
        self._x = x
        self._prepare_weight_images(P, train)
        region = self._region_begin()         # bench hook: the `inc` double-conv forward (2 convs, BN statistics + apply)
        if not self.first_direct:
            dcol = L.make_nhwc(self.col0)
            check(lib.gsd_bf16_im2col3x3(x.data_ptr(), n, c, h, w, C.byref(dcol), st), "im2col3x3")
        for lvl in range(self.L + 1):
            u0, u1 = self.enc[lvl]
            if lvl == 1:
                self._region_end("inc_forward", region)
            if lvl == 0:
                src = (None, 0, 0) if self.first_direct else (self.col0, 0, self.col0.shape[3])
            else:
                prev = self.enc[lvl - 1][1]
                if not (train and self.apply_pool):     # (train mode: the pool rode in the previous unit's BatchNorm apply)
                    dp = L.make_nhwc(self.pooled[lvl])
                    check(lib.gsd_bf16_maxpool2(C.byref(prev.a), C.byref(dp), st), "maxpool2")
                src = (self.pooled[lvl], 0, prev.cout)
            pool_to = self.pooled[lvl + 1] if (train and self.apply_pool and lvl < self.L) else None
            if lvl == 0 and train and self.fused_inc:
                self._run_inc_fused(u0, u1, P, st, pool_to)
                continue
            self._run_unit(u0, src, P, train, st)
            self._run_unit(u1, (u0.a_t, u0.a_off, u0.cout), P, train, st, pool_to=pool_to)
        if self.L == 0:
            self._region_end("inc_forward", region)
        cur = self.enc[self.L][1]
        z = L.int_array([0])
        for j in range(self.L):
            up = self.ups[j]
            lvl = self.L - 1 - j
            self._wimage(3, P[up.wname], up.cout, up.cin, up.wt_f, st)
            oy, ox = self._pad_off(lvl)
            dslice = L.make_nhwc(self.cat[lvl], self.dims[lvl], up.cout)
            done = self._log("bf16_convT", 2.0 * 4 * up.cout * up.cin * n * self.hs[lvl + 1] * self.ws[lvl + 1])
            check(lib.gsd_bf16_conv_dense(C.byref(cur.a), up.wt_f.data_ptr(), C.byref(dslice), up.cin, 4 * up.cout, 1, 1, z, z,
                                          self.hs[lvl + 1], self.ws[lvl + 1], up.cout, oy, ox, P[up.bname].data_ptr(), None, None, st),
                  "convT")
            done()
            u0, u1 = self.dec[j]
            self._run_unit(u0, (self.cat[lvl], 0, u0.cin), P, train, st)
            self._run_unit(u1, (u0.a_t, u0.a_off, u0.cout), P, train, st)
            cur = u1
        if self._nbt:
            L.add_counters(self._nbt, 1)
            self._nbt = []
        if out is None:
            out = torch.empty((n, self.n_classes, h, w), device=x.device, dtype=torch.float32)
        if train and self.fused_out:
            check(lib.gsd_bf16_bn_relu_conv1x1_out(C.byref(L.make_nhwc(cur.y)), cur.scale.data_ptr(), cur.shift.data_ptr(),
                                                   P["outc.conv.weight"].data_ptr(), P["outc.conv.bias"].data_ptr(), self.n_classes,
                                                   out.data_ptr(), st), "bn_relu_conv1x1_out")
        else:
            check(lib.gsd_bf16_conv1x1_out(C.byref(cur.a), P["outc.conv.weight"].data_ptr(), P["outc.conv.bias"].data_ptr(),
                                           self.n_classes, out.data_ptr(), st), "conv1x1_out")
        return out

    # ------------------------------------------------------------------ backward
    def _reduce(self, mode: int, u: _Unit, st: int, g: Optional[L.gsd_nhwc] = None, dpool: Optional[torch.Tensor] = None,
                dout: Optional[torch.Tensor] = None, wout: Optional[torch.Tensor] = None) -> None:
        """pass 1 of BatchNorm+ReLU backward: dz -> u.g, per-block sums -> self.partials."""
        dy, dz = L.make_nhwc(u.y), L.make_nhwc(u.g)
        gsrc = g if g is not None else dz
        dp = L.make_nhwc(dpool) if dpool is not None else dz
        check(lib.gsd_bf16_bn_bwd_reduce(mode, C.byref(dy), u.scale.data_ptr(), u.shift.data_ptr(), u.mean.data_ptr(),
                                         u.invstd.data_ptr(), C.byref(gsrc), C.byref(u.a), C.byref(dp), L.ptr(dout), L.ptr(wout),
                                         C.byref(dz), self.partials.data_ptr(), st), "bn_bwd_reduce")

    def _tail(self, u: _Unit, G, st: int, dwout: Optional[torch.Tensor] = None, fused: bool = False, recompute: bool = False) -> None:
        """u.g holds dz and self.partials its sums: finish BatchNorm backward (dgamma, dbeta, dy in place), then dW.
        fused: the sums come from a dX launch's epilogue (conv partial layout) instead of gsd_bf16_bn_bwd_reduce.
        recompute (the fused `inc`'s first unit): u.g holds da, the gradient w.r.t. the unit's ACTIVATION, and its raw output
        does not exist: pass 1 and dW recompute it from x (gsd_bf16_first_bn_bwd_reduce, gsd_bf16_wgrad_first_recompute)."""
        n, lh, lw = u.y.shape[0], self.hs[u.level], self.ws[u.level]
        count = float(n * lh * lw)
        rows = (u.fused_rows or lib.gsd_bf16_conv_partial_rows(n, lh, lw, u.cout)) if fused else lib.gsd_bf16_bn_bwd_partial_rows(n, lh, lw)
        if recompute:
            fused = True
            rows = lib.gsd_bf16_conv3x3_first_partial_rows(n, lh, lw, u.cout)
            da = L.make_nhwc(u.g)
            check(lib.gsd_bf16_first_bn_bwd_reduce(self._x.data_ptr(), n, u.cin, lh, lw, u.wt_f.data_ptr(), C.byref(da), u.scale.data_ptr(),
                                                   u.shift.data_ptr(), u.mean.data_ptr(), u.invstd.data_ptr(), self.partials.data_ptr(), st),
                  "first_bn_bwd_reduce")
        dw_ptr = None if dwout is None else dwout.data_ptr()
        if self.sync_fn is None:
            check(lib.gsd_bn_bwd_reduce_finalize(self.partials.data_ptr(), rows, lib.gsd_bf16_conv_mpad(u.cout) if fused else 0, u.cout,
                                                 u.sums.data_ptr(), count, G[u.gname].data_ptr(), G[u.bname].data_ptr(), dw_ptr,
                                                 u.c1.data_ptr(), u.c2.data_ptr(), st), "bn_bwd_reduce_finalize")
        else:
            if fused:
                check(lib.gsd_bn_reduce_partials(self.partials.data_ptr(), rows, lib.gsd_bf16_conv_mpad(u.cout), u.cout,
                                                 u.sums.data_ptr(), st), "bn_reduce_partials")
            else:
                check(lib.gsd_bn_bwd_reduce_partials(self.partials.data_ptr(), rows, u.cout, u.sums.data_ptr(), st),
                      "bn_bwd_reduce_partials")
            gsum = u.sums[:2 * u.cout].clone()
            self.sync_fn(gsum)
            count *= self.world
            check(lib.gsd_bn_bwd_finalize(u.sums.data_ptr(), gsum.data_ptr(), u.cout, count, G[u.gname].data_ptr(),
                                          G[u.bname].data_ptr(), dw_ptr, u.c1.data_ptr(), u.c2.data_ptr(), st), "bn_bwd_finalize")
        dz, dy = L.make_nhwc(u.g), L.make_nhwc(u.y)
        if recompute:
            def launch(sst, ws):
                done = self._log("bf16_wgrad", 2.0 * u.cout * 9 * u.cin * n * lh * lw)
                check(lib.gsd_bf16_wgrad_first_recompute(self._x.data_ptr(), n, u.cin, lh, lw, u.wt_f.data_ptr(), C.byref(dz),
                                                         u.scale.data_ptr(), u.shift.data_ptr(), u.mean.data_ptr(), u.invstd.data_ptr(),
                                                         u.c1.data_ptr(), u.c2.data_ptr(), G[u.wname].data_ptr(), ws.data_ptr(), ws.numel(),
                                                         sst), "wgrad_first_recompute")
                done()
            self._on_side(launch)
            return
        if u.first and self.first_direct:
            # no dX for the first layer, so dW is d_raw's only reader: it forms d_raw from (dz, y) itself -- no apply pass, no im2col
            def launch(sst, ws):
                done = self._log("bf16_wgrad", 2.0 * u.cout * 9 * u.cin * n * lh * lw)
                check(lib.gsd_bf16_wgrad_first(self._x.data_ptr(), n, u.cin, lh, lw, C.byref(dz), C.byref(dy), u.scale.data_ptr(),
                                               u.mean.data_ptr(), u.invstd.data_ptr(), u.c1.data_ptr(), u.c2.data_ptr(),
                                               G[u.wname].data_ptr(), ws.data_ptr(), ws.numel(), sst), "wgrad_first")
                done()
            self._on_side(launch)
            return
        check(lib.gsd_bf16_bn_bwd_apply(C.byref(dz), C.byref(dy), u.scale.data_ptr(), u.mean.data_ptr(), u.invstd.data_ptr(),
                                        u.c1.data_ptr(), u.c2.data_ptr(), st), "bn_bwd_apply")
        db = L.make_nhwc(*u.src)

        def launch(sst, ws):
            if u.first:
                z = L.int_array([0])
                done = self._log("bf16_wgrad", 2.0 * u.cout * u.src[2] * n * lh * lw)
                check(lib.gsd_bf16_wgrad(C.byref(dz), C.byref(db), 1, 1, z, z, G[u.wname].data_ptr(), 9 * u.cin, ws.data_ptr(),
                                         ws.numel(), sst), "wgrad(first)")
            else:
                done = self._log("bf16_wgrad", 2.0 * u.cout * u.cin * 9 * n * lh * lw)
                check(lib.gsd_bf16_wgrad(C.byref(dz), C.byref(db), 9, 1, L.int_array(T3Y), L.int_array(T3X), G[u.wname].data_ptr(),
                                         u.cin, ws.data_ptr(), ws.numel(), sst), "wgrad")
            done()
        self._on_side(launch)

    def _image_jobs(self, P, train: bool):
        """(mode, weights, cout, cin, image buffer) of every weight image a step reads: the forward images, and in train mode
        the dX images of the units that have a dX."""
        jobs = []
        for pair in self.enc:
            for u in pair:
                jobs.append((2 if u.first else 0, P[u.wname], u.cout, u.cin, u.wt_f))
        for j in range(self.L):
            up = self.ups[j]
            jobs.append((3, P[up.wname], up.cout, up.cin, up.wt_f))
            for u in self.dec[j]:
                jobs.append((0, P[u.wname], u.cout, u.cin, u.wt_f))
        if train:
            for j in range(self.L):
                for u in self.dec[j]:
                    jobs.append((1, P[u.wname], u.cout, u.cin, u.wt_d))
                jobs.append((4, P[self.ups[j].wname], self.ups[j].cout, self.ups[j].cin, self.ups[j].wt_d))
            for pair in self.enc:
                for u in pair:
                    if u.need_dgrad:
                        jobs.append((1, P[u.wname], u.cout, u.cin, u.wt_d))
        return jobs

    def _prepare_weight_images(self, P, train: bool) -> None:
        """Every bf16 weight image of the step in ONE launch per 32 images at the start of the forward (gsd_bf16_weight_images):
        they depend on nothing but the parameters, and as 43 latency-bound launches between the convolutions they cost 0.45 ms
        of a 30-ms step.  `_wimage` then finds the image done.  GSD_BF16_BATCH_WIMG=0: one launch per image, where it is used.
        (GSD_BF16_SIDE_WIMG=1, the per-image launches on the side stream instead, measured slower: 30.7 vs 30.4 ms.)"""
        self._wready = {}
        self._wdone = set()
        if os.environ.get("GSD_BF16_BATCH_WIMG", "1") != "0":
            jobs = self._image_jobs(P, train)
            arr = (L.gsd_bf16_wimg_job * len(jobs))()
            for i, (mode, w, cout, cin, buf) in enumerate(jobs):
                arr[i].w, arr[i].out, arr[i].mode, arr[i].Cout, arr[i].Cin = w.data_ptr(), buf.data_ptr(), mode, cout, cin
                self._wdone.add(buf.data_ptr())
            check(lib.gsd_bf16_weight_images(arr, len(jobs), L.stream_ptr()), "weight_images")
            return
        if not (train and self.side_dw and os.environ.get("GSD_BF16_SIDE_WIMG", "0") == "1"):
            return
        self.side.wait_stream(torch.cuda.current_stream())     # the previous step's Adam update, and its last readers of the images
        with torch.cuda.stream(self.side):
            sst = L.stream_ptr()
            for mode, w, cout, cin, buf in self._image_jobs(P, train):
                check(lib.gsd_bf16_weight_image(mode, w.data_ptr(), cout, cin, buf.data_ptr(), sst), "weight_image")
                ev = self._wevents.get(buf.data_ptr())
                if ev is None:
                    ev = self._wevents[buf.data_ptr()] = torch.cuda.Event()
                ev.record()
                self._wready[buf.data_ptr()] = ev

    def _wimage(self, mode: int, w: torch.Tensor, cout: int, cin: int, buf: torch.Tensor, st: int) -> None:
        if buf.data_ptr() in self._wdone:                     # produced by the batched launch at the start of this step
            return
        ev = self._wready.pop(buf.data_ptr(), None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)        # produced on the side stream at the start of this step
            return
        check(lib.gsd_bf16_weight_image(mode, w.data_ptr(), cout, cin, buf.data_ptr(), st), "weight_image")

    def _on_side(self, launch) -> None:
        """Run launch(stream pointer, workspace tensor) -- one weight-gradient launch -- behind everything issued so far, on the
        side stream when there is one."""
        if not self.side_dw or self.kernel_log is not None:   # (a per-kernel timing pass wants every kernel alone on the chip)
            launch(L.stream_ptr(), self.wspace)
            return
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            launch(L.stream_ptr(), self.wspace_side)

    def _join_side(self) -> None:
        if self.side_dw:
            torch.cuda.current_stream().wait_stream(self.side)

    def _announce(self, tag: str) -> None:
        """As engine.UNetEngine._announce: the bucket's all-reduce is called under a hand-off stream that waits for the main and
        the side stream; the main stream (the dX chain) does not wait for the weight gradients."""
        if self.block_done_cb is None:
            return
        if not self.side_dw or self.kernel_log is not None:
            self.block_done_cb(tag)
            return
        if getattr(self, "_handoff", None) is None:
            self._handoff = torch.cuda.Stream(device=self.side.device)
        self._handoff.wait_stream(torch.cuda.current_stream())
        self._handoff.wait_stream(self.side)
        with torch.cuda.stream(self._handoff):
            self.block_done_cb(tag)

    def _bnbwd(self, tgt: _Unit):
        """gsd_bf16_bnbwd for fusing pass 1 of tgt's BatchNorm+ReLU backward into the dX launch that produces tgt.g."""
        yv = L.make_nhwc(tgt.y)
        bw = L.gsd_bf16_bnbwd()
        bw.y = C.pointer(yv)
        bw.scale, bw.shift, bw.mean, bw.invstd = (tgt.scale.data_ptr(), tgt.shift.data_ptr(), tgt.mean.data_ptr(),
                                                  tgt.invstd.data_ptr())
        return bw, yv

    def _dgrad(self, u: _Unit, P, dst: torch.Tensor, st: int, fuse: Optional[_Unit] = None, stats_to: Optional[torch.Tensor] = None
               ) -> None:
        """dX of unit u (u.g holds dy) into the plain tensor dst (N,H,W,u.cin); with `fuse` (the unit whose activation is
        u's input, dst == fuse.g) the epilogue also does pass 1 of that unit's BatchNorm+ReLU backward; with `stats_to` (and no
        fuse) it leaves the per-channel sums of what it stores there (gsd_bf16_conv_partial_rows rows of 2*mpad(u.cin))."""
        n, lh, lw = u.y.shape[0], self.hs[u.level], self.ws[u.level]
        self._wimage(1, P[u.wname], u.cout, u.cin, u.wt_d, st)
        din, dout = L.make_nhwc(u.g), L.make_nhwc(dst)
        bw = keep = None
        if fuse is not None:
            bw, keep = self._bnbwd(fuse)
        done = self._log("bf16_conv3x3", 2.0 * u.cout * u.cin * 9 * n * lh * lw, (u.cout, u.cin, lh, lw))
        part = self.partials.data_ptr() if bw is not None else (stats_to.data_ptr() if stats_to is not None else None)
        if self._use_c64(u.cout, u.cin):
            check(lib.gsd_bf16_conv3x3_c64(C.byref(din), u.wt_d.data_ptr(), C.byref(dout), part, C.byref(bw) if bw is not None else None, st),
                  "conv3x3_c64 dgrad")
            rows = lib.gsd_bf16_conv3x3_c64_partial_rows(n, lh, lw)
        else:
            check(lib.gsd_bf16_conv3x3(C.byref(din), u.wt_d.data_ptr(), C.byref(dout), u.cout, u.cin, part,
                                       C.byref(bw) if bw is not None else None, st), "conv3x3 dgrad")
            rows = lib.gsd_bf16_conv_partial_rows(n, lh, lw, u.cin)
        if fuse is not None:
            fuse.fused_rows = rows
        done()

    def backward(self, dout: torch.Tensor, P: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor]) -> None:
        if not self._saved_train:
            raise L.GsdError("backward() needs a preceding train-mode forward()")
        dout = dout.contiguous()
        st = L.stream_ptr()
        n = dout.shape[0]
        last = self.dec[-1][1] if self.L > 0 else self.enc[0][1]
        self._reduce(2, last, st, dout=dout, wout=P["outc.conv.weight"])
        check(lib.gsd_sum_planes(dout.data_ptr(), n, self.n_classes, dout.shape[2] * dout.shape[3], G["outc.conv.bias"].data_ptr(),
                                 self.wspace.data_ptr(), st), "sum_planes")
        dwout = G["outc.conv.weight"]
        prev_fused = None           # unit whose pass-1 sums came from the transposed convolution's dX epilogue
        for j in reversed(range(self.L)):
            u0, u1 = self.dec[j]
            up = self.ups[j]
            lvl = self.L - 1 - j
            self._tail(u1, G, st, dwout, fused=prev_fused is u1)
            dwout = None
            self._dgrad(u1, P, u0.g, st, fuse=u0)
            self._tail(u0, G, st, fused=True)
            self._dgrad(u0, P, self.gcat[lvl], st, stats_to=self.db_part[lvl] if self.db_from_dx else None)
            prev = self.dec[j - 1][1] if j > 0 else self.enc[self.L][1]
            hi, wi = self.hs[lvl + 1], self.ws[lvl + 1]
            oy, ox = self._pad_off(lvl)
            gup = L.make_nhwc(self.gcat[lvl], self.dims[lvl], up.cout)
            ty, tx = L.int_array([oy, oy, oy + 1, oy + 1]), L.int_array([ox, ox + 1, ox, ox + 1])
            def launch(sst, ws, up=up, prev=prev, gup=gup, ty=ty, tx=tx, hi=hi, wi=wi, oy=oy, ox=ox, lvl=lvl):
                done = self._log("bf16_wgrad", 2.0 * 4 * up.cout * up.cin * n * hi * wi)
                check(lib.gsd_bf16_wgrad(C.byref(prev.a), C.byref(gup), 4, 2, ty, tx, G[up.wname].data_ptr(), up.cout, ws.data_ptr(),
                                         ws.numel(), sst), "convT wgrad")
                done()
                # (the bias gradient -- per-channel sums of the same gradient slice -- is off the critical path too)
                if self.db_from_dx:      # from the statistics rows the dX launch above left: no pass over the slice
                    ctot = self.cat[lvl].shape[3]
                    check(lib.gsd_bf16_convT_bias_grad(self.db_part[lvl].data_ptr(), lib.gsd_bf16_conv_partial_rows(n, self.hs[lvl], self.ws[lvl], ctot),
                                                       2 * lib.gsd_bf16_conv_mpad(ctot), ctot - up.cout, C.byref(gup), oy, ox, 2 * hi, 2 * wi,
                                                       G[up.bname].data_ptr(), ws.data_ptr(), ws.numel(), sst), "convT bias grad")
                else:
                    check(lib.gsd_bf16_channel_sums(C.byref(gup), oy, ox, 2 * hi, 2 * wi, G[up.bname].data_ptr(), ws.data_ptr(),
                                                    ws.numel(), sst), "convT bias grad")
            self._on_side(launch)
            self._wimage(4, P[up.wname], up.cout, up.cin, up.wt_d, st)
            dprev = L.make_nhwc(prev.g)
            done = self._log("bf16_convT", 2.0 * 4 * up.cout * up.cin * n * hi * wi)
            bw, keep = self._bnbwd(prev)
            check(lib.gsd_bf16_conv_dense(C.byref(gup), up.wt_d.data_ptr(), C.byref(dprev), up.cout, up.cin, 4, 2, ty, tx, hi, wi, 0,
                                          0, 0, None, self.partials.data_ptr(), C.byref(bw), st), "convT dgrad")
            done()
            prev.fused_rows = lib.gsd_bf16_conv_dense_partial_rows(n, hi, wi, up.cout, up.cin, 4, 2)
            prev_fused = prev
            self._announce(f"dec{j}")
        for lvl in reversed(range(self.L + 1)):
            u0, u1 = self.enc[lvl]
            if lvl < self.L:
                gskip = L.make_nhwc(self.gcat[lvl], 0, u1.cout)
                if self.pool_index:
                    dyv, dzv, dpv = L.make_nhwc(u1.y), L.make_nhwc(u1.g), L.make_nhwc(self.dpooled[lvl + 1])
                    check(lib.gsd_bf16_bn_bwd_reduce_pool_idx(C.byref(dyv), u1.scale.data_ptr(), u1.shift.data_ptr(), u1.mean.data_ptr(),
                                                              u1.invstd.data_ptr(), C.byref(gskip), self.pool_idx[lvl + 1].data_ptr(),
                                                              C.byref(dpv), C.byref(dzv), self.partials.data_ptr(), st),
                          "bn_bwd_reduce_pool_idx")
                else:
                    self._reduce(1, u1, st, g=gskip, dpool=self.dpooled[lvl + 1])
            self._tail(u1, G, st, dwout, fused=prev_fused is u1)
            dwout = None
            inc = lvl == 0 and self.fused_inc      # u0's raw output was never stored: no pass 1 in the dX epilogue
            self._dgrad(u1, P, u0.g, st, fuse=None if inc else u0)
            self._tail(u0, G, st, fused=True, recompute=inc)
            self._announce(f"enc{lvl}")
            if lvl > 0:
                self._dgrad(u0, P, self.dpooled[lvl], st)
        self._join_side()
