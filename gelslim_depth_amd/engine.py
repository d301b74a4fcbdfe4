"""Host-side schedule of the U-Net hot path over libgsd's C ABI.

Mirrors the control flow of the reference model
(/root/reference/gelslim_depth/models/unet.py:79-88 forward; autograd's reverse sweep for backward)
but every arithmetic step is a libgsd kernel launch on torch's current HIP stream.  torch is used
for device memory (buffers) and, in data-parallel runs, for the RCCL collectives; no torch op
computes any part of the path (tests/test_gpu_robust.py checks that a fused train step dispatches no ATen compute op).

Data layout in HBM (all fp32 NCHW):
  raw[u]    raw conv3x3 output of every conv unit (pre-BN); the normalised/activated tensor is
            never stored, consumers apply (scale, shift, relu) on load            -- 18 tensors
  g[u]      gradient buffer of the same shape: da -> dz -> d_raw in place         -- 18 tensors
  pooled[l] max-pool output feeding encoder level l (l>=1);  dpooled[l] its gradient
  up[j]     transposed-conv output (+bias) of decoder j at (2h,2w), unpadded;  dup[j] its gradient
  wt_*      per-step re-laid-out weights (k-major, out-channel contiguous)
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


def _r64(c: int) -> int:
    return (c + 63) // 64 * 64


def _r4(w: int) -> int:
    return (w + 3) // 4 * 4


class _ConvForm:
    """Entry points and weight-layout modes of one of the three conv3x3 forms: 0 direct taps, 1 Winograd F(4,3) along rows,
    2 two-dimensional Winograd F(2x4,3x3)."""

    def __init__(self, algo: int):
        self.algo = algo
        self.conv = (lib.gsd_conv3x3, lib.gsd_conv3x3_w43, lib.gsd_conv3x3_w2d)[algo]
        self.dgrad_bnrelu = (lib.gsd_conv3x3_dgrad_bnrelu, lib.gsd_conv3x3_w43_dgrad_bnrelu, lib.gsd_conv3x3_w2d_dgrad_bnrelu)[algo]
        self.partial_rows = (lib.gsd_conv3x3_partial_rows, lib.gsd_conv3x3_w43_partial_rows, lib.gsd_conv3x3_w2d_partial_rows)[algo]
        self.mode_f, self.mode_d = ((0, 1), (4, 5), (8, 9))[algo]
        # K-slab scratch the form would like for a shape (the two Winograd forms; 0: the shape runs unsplit)
        self.workspace = (lambda *a: 0, lib.gsd_conv3x3_w43_workspace, lib.gsd_conv3x3_w2d_workspace)[algo]

    @staticmethod
    def choose(n: int, h: int, w: int, cin: int, c0: int, cout: int, train: bool) -> "_ConvForm":
        """The library's preference for a launch shape (include/gsd.h: gsd_conv3x3_algo, gsd_conv3x3_prefers_w2d).  c0: channels
        of the first source segment.  Eval mode gets the forms whose bits do not depend on the batch (no row folding, no K slabs)."""
        if not train:
            # decided from N-independent quantities only: the one-image plan says direct or Winograd; Winograd means the 2-D form (no
            # row folding, no K slabs: image i of a batch gets the bits the image alone gets); where the 2-D form does not serve
            # the channel counts the direct form does (its tiles never span images either) -- never the row form, which folds rows
            # across images
            algo = lib.gsd_conv3x3_algo(1, h, w, cin, cout)
            if algo == 1:
                algo = 2 if lib.gsd_conv3x3_w2d_supported(cin, c0) else 0
            return _ConvForm(algo)
        algo = lib.gsd_conv3x3_algo(n, h, w, cin, cout)
        if algo == 1 and lib.gsd_conv3x3_w2d_supported(cin, c0) and lib.gsd_conv3x3_prefers_w2d(n, h, w, cin, cout, int(train)):
            algo = 2
        return _ConvForm(algo)

    def run(self, ws, src, nsrc, wt, cin, cout, dst, ndst, part, n, h, w, st):
        """conv3x3 forward / dX; `ws`: the engine's K-slab scratch (train mode) or None."""
        if self.algo >= 1 and ws is not None:
            fn = lib.gsd_conv3x3_w43_ws if self.algo == 1 else lib.gsd_conv3x3_w2d_ws
            return fn(src, nsrc, wt, cin, cout, dst, ndst, part, ws.data_ptr(), ws.numel(), n, h, w, st)
        return self.conv(src, nsrc, wt, cin, cout, dst, ndst, part, n, h, w, st)

    def run_bnrelu(self, ws, src, wt, cin, cout, dst, raw, scale, shift, mean, invstd, part, n, h, w, st):
        if self.algo >= 1 and ws is not None:
            fn = lib.gsd_conv3x3_w43_dgrad_bnrelu_ws if self.algo == 1 else lib.gsd_conv3x3_w2d_dgrad_bnrelu_ws
            return fn(src, wt, cin, cout, dst, raw, scale, shift, mean, invstd, part, ws.data_ptr(), ws.numel(), n, h, w, st)
        return self.dgrad_bnrelu(src, wt, cin, cout, dst, raw, scale, shift, mean, invstd, part, n, h, w, st)


class _Unit:
    """conv3x3(no bias) + BatchNorm2d + ReLU (unet.py:11-13 / :14-16)."""

    def __init__(self, prefix: str, conv_idx: int, bn_idx: int, cin: int, cout: int, level: int):
        self.prefix, self.conv_idx, self.bn_idx = prefix, conv_idx, bn_idx
        self.cin, self.cout, self.level = cin, cout, level
        self.wname = f"{prefix}.double_conv.{conv_idx}.weight"
        bn = f"{prefix}.double_conv.{bn_idx}."
        self.gname, self.bname = bn + "weight", bn + "bias"
        self.rmname, self.rvname, self.nbtname = bn + "running_mean", bn + "running_var", bn + "num_batches_tracked"
        self.need_dgrad = True
        self.c0 = cin              # channels of the first source segment (the decoder's first convs: the skip tensor's)
        # device buffers (filled by the engine)
        self.wt_f = self.wt_d = None
        self.scale = self.shift = self.mean = self.invstd = self.c1 = self.c2 = None
        self.sums = None
        self.raw = self.g = None
        self.dsrc = None  # d_raw as the dW / dX kernels read it: g itself, or the pitched scratch buffer (engine.gp)
        self.pitched = False   # gsd_bn_bwd_apply writes d_raw out of place into the pitched buffer
        self.fused_dw = False  # first layer: no dX, so dW forms d_raw itself (gsd_conv3x3_wgrad_bn) and the apply pass is skipped
        self.srcs = None  # gsd_src array kept for wgrad
        self.form_f = self.form_d = None   # _ConvForm of the forward / dX launch for the current shape
        self.forms_f = None                # ... of the forward launch in eval (False) and train (True) mode
        self.fused_rows = 0                # partial rows written by the dX launch that produced this unit's dz


class _Up:
    """ConvTranspose2d(cin, cin//2, 2, 2) (unet.py:36)."""

    def __init__(self, j: int, cin: int, level_in: int):
        self.j, self.cin, self.cout, self.level_in = j, cin, cin // 2, level_in
        self.wname, self.bname = f"up.{j}.up.weight", f"up.{j}.up.bias"
        self.wt_f = self.wt_d = None
        self.mode_d = 3            # gsd_weight_layout mode of wt_d (gsd_convT2x2_dgrad_layout)
        self.bn_rows = 0           # > 0: the dX launch also does pass 1 of the BatchNorm backward of the unit below (partial rows)
        self.out = self.dout = None


class UNetEngine:
    def __init__(self, n_channels: int, n_classes: int, layer_dimensions: Sequence[int]):
        dims = list(layer_dimensions)
        self.n_channels, self.n_classes, self.dims = n_channels, n_classes, dims
        self.L = len(dims) - 1
        self.enc: List[Tuple[_Unit, _Unit]] = []
        self.dec: List[Tuple[_Unit, _Unit]] = []
        self.ups: List[_Up] = []
        self.enc.append((_Unit("inc", 0, 1, n_channels, dims[0], 0), _Unit("inc", 3, 4, dims[0], dims[0], 0)))
        self.enc[0][0].need_dgrad = False
        for i in range(self.L):
            p = f"down.{i}.maxpool_conv.1"
            self.enc.append((_Unit(p, 0, 1, dims[i], dims[i + 1], i + 1), _Unit(p, 3, 4, dims[i + 1], dims[i + 1], i + 1)))
        for j, i in enumerate(range(self.L, 0, -1)):
            cin, cout = dims[i], dims[i - 1]
            # Up(cin, cout): ConvTranspose2d(cin, cin//2) then DoubleConv(cin, cout) on cat[skip, up] (unet.py:36-37,48):
            # the reference only runs when skip channels + cin//2 == cin.
            if dims[i - 1] + cin // 2 != cin:
                raise ValueError(f"layer_dimensions {dims}: level {i} needs dims[i-1] + dims[i]//2 == dims[i] "
                                 "(the reference model fails at torch.cat/conv otherwise)")
            self.ups.append(_Up(j, cin, i))
            p = f"up.{j}.conv"
            # DoubleConv(in_channels=cin, out=cout): input = cat[skip (dims[i-1]), up (cin//2)]
            self.dec.append((_Unit(p, 0, 1, cin, cout, i - 1), _Unit(p, 3, 4, cout, cout, i - 1)))
            self.dec[-1][0].c0 = dims[i - 1]
        self.units: List[_Unit] = [u for pair in self.enc for u in pair] + [u for pair in self.dec for u in pair]
        self._shape = None
        self._dev = None
        self.sync_fn: Optional[Callable[[torch.Tensor], None]] = None   # SyncBN hook: all-reduce fp64 sums in place
        self.world = 1
        self._saved_train = False
        self.block_done_cb: Optional[Callable[[str], None]] = None   # data-parallel hook: a block's grads are final
        self._nbt: list = []   # num_batches_tracked buffers of the current train-mode forward
        self.guard = None          # non-finite guard of the current step (_lib.make_guard), set by TrainStep per step
        self.generation = 0        # forwards so far: a backward belongs to exactly one (models/unet.py checks it)
        # bench hook: when a list, every conv3x3 launch appends (variant, flops, start_event, end_event)
        # partial-row count up to which BatchNorm's column sums and finalize run as ONE launch (gsd_bn_[bwd_]reduce_finalize)
        self.one_launch_rows = int(os.environ.get("GSD_BN_ONE_LAUNCH_ROWS", "4096"))
        # weight gradients on a side stream: dW of a unit runs beside its dX and the BatchNorm backward of the unit below (they only
        # share d_raw as an input).  A dW block owns its CU (8 waves x 240 registers, 96 KiB of LDS), so the two streams interleave CU
        # by CU: dX keeps its stand-alone speed and dW fills the CUs dX's tails and the chain's small launches leave idle
        self.side_dw = os.environ.get("GSD_SIDE_DW", "1") != "0"
        self.side: Optional[torch.cuda.Stream] = None
        self.convt_dg_bn = os.environ.get("GSD_CONVT_DG_BN", "1") != "0"   # ConvT dX + pass 1 of the BatchNorm backward below it
        self.batch_wl = os.environ.get("GSD_WL_BATCH", "1") != "0"   # a pass's weight layouts through gsd_weight_layout_batch
        self._handoff: Optional[torch.cuda.Stream] = None   # bucket hand-off to the all-reduce (see _announce)
        self.kernel_log: Optional[list] = None
        self.wgrad_log: Optional[list] = None      # bench hook: conv3x3 dW launches (+ slab reducer), same rows as kernel_log
        self.region_log: Optional[list] = None     # bench hook: (region name, start event, end event)

    # ------------------------------------------------------------------ buffers
    # Library switches that change launch plans (tile shapes, partial-row counts, slab counts, the form a launch takes): partials,
    # conv_ws and the weight-gradient workspaces are sized from them, so they are part of the shape key -- a switch flipped
    # between two steps (a test's monkeypatch, a sweep in one process) re-sizes the buffers instead of overrunning them
    _SIZING_ENV = ("GSD_W2D_WAVES", "GSD_W2D_TW", "GSD_W2D_TW8_PCT", "GSD_W43_TW", "GSD_W43_FOLD", "GSD_CONV_W2D", "GSD_CONV_ALGO",
                   "GSD_W43_SPLIT", "GSD_WGRAD_ALGO", "GSD_WGRAD_W2D", "GSD_WG2D_KX", "GSD_WG2D_BLOCKS", "GSD_WGRAD_BLOCKS",
                   "GSD_WG43_TW", "GSD_WG43_SMALL")

    def _ensure(self, n: int, h: int, w: int, dev: torch.device, train: bool) -> None:
        key = (n, h, w, str(dev), tuple(os.environ.get(k) for k in self._SIZING_ENV))
        if self._shape == key and (not train or self.units[0].g is not None):
            return
        if self._shape != key:
            for u in self.units:
                u.raw = u.g = None
            for up in self.ups:
                up.out = up.dout = None
        self._shape = key
        self._dev = dev
        hs, ws = [h], [w]
        for _ in range(self.L):
            hs.append(hs[-1] // 2)
            ws.append(ws[-1] // 2)
        assert hs[-1] >= 1 and ws[-1] >= 1, "input too small for this many max-pools"
        self.hs, self.ws = hs, ws
        f32 = dict(device=dev, dtype=torch.float32)
        max_part = 1
        max_ws = 1
        max_gp = 0
        max_slab = 0
        for u in self.units:
            lh, lw = hs[u.level], ws[u.level]
            if u.raw is None:
                u.raw = L.slack_empty((n, u.cout, lh, lw), dev)   # dW reads it as 16-byte window pieces
            if train and u.g is None:
                u.g = torch.empty((n, u.cout, lh, lw), **f32)
            if u.scale is None or u.scale.device != dev:
                for nm in ("scale", "shift", "mean", "invstd", "c1", "c2"):
                    setattr(u, nm, torch.empty((u.cout,), **f32))
                u.sums = torch.empty((65 * 3 * u.cout,), device=dev, dtype=torch.float64)
            # direct taps or Winograd F(4,3) rows, per layer shape and per direction (include/gsd.h: gsd_conv3x3_algo)
            # direct taps, Winograd F(4,3) rows or two-dimensional Winograd, per layer shape, per direction and per mode: an
            # eval-mode forward takes the forms whose bits do not depend on the batch size (_ConvForm.choose)
            u.forms_f = {t: _ConvForm.choose(n, lh, lw, u.cin, u.c0, u.cout, t) for t in (False, True)}
            u.form_f = u.forms_f[train]
            u.form_d = _ConvForm.choose(n, lh, lw, u.cout, u.cout, u.cin, True) if u.need_dgrad else None
            need = max(lib.gsd_weight_layout_size(f.mode_f, u.cout, u.cin) for f in u.forms_f.values())
            if u.wt_f is None or u.wt_f.numel() != need or u.wt_f.device != dev:
                u.wt_f = torch.empty((need,), **f32)
            if u.need_dgrad:
                need = lib.gsd_weight_layout_size(u.form_d.mode_d, u.cout, u.cin)
                if u.wt_d is None or u.wt_d.numel() != need or u.wt_d.device != dev:
                    u.wt_d = torch.empty((need,), **f32)
                max_part = max(max_part, u.form_d.partial_rows(n, lh, lw, u.cin) * 2 * _r64(u.cin))
            rows = max(f.partial_rows(n, lh, lw, u.cout) for f in u.forms_f.values())
            max_part = max(max_part, rows * 2 * _r64(u.cout))
            if train:
                # K slabs (gsd_conv3x3_w43_ws) for the launches that would leave most of the chip idle: train mode only --
                # an eval-mode forward keeps the one summation order whatever the batch (image i of a batch == the image alone)
                max_slab = max(max_slab, u.form_f.workspace(n, lh, lw, u.cin, u.cout),
                               u.form_d.workspace(n, lh, lw, u.cout, u.cin) if u.need_dgrad else 0)
                max_part = max(max_part, lib.gsd_bn_bwd_partial_rows(n, u.cout, lh, lw) * 3 * u.cout)
                max_ws = max(max_ws, lib.gsd_conv3x3_wgrad_workspace(n, lh, lw, u.cin, u.cout))
                # d_raw goes to a row-pitched scratch buffer (16-byte aligned rows) when both of its readers -- dW and dX
                # of this unit -- are the Winograd kernels, which then move it as aligned 16-byte LDS-DMA pieces
                u.pitched = bool(lib.gsd_conv3x3_wgrad_takes_pitched_dy(n, lh, lw, u.cin, u.cout)) and \
                    (not u.need_dgrad or u.form_d.algo >= 1)
                if u.pitched:
                    max_gp = max(max_gp, n * u.cout * lh * _r4(lw))
                u.fused_dw = (not u.need_dgrad) and bool(lib.gsd_conv3x3_wgrad_bn_supported(n, lh, lw, u.cin, u.cout))
                if u.fused_dw:
                    max_ws = max(max_ws, lib.gsd_conv3x3_wgrad_bn_workspace(n, lh, lw, u.cin, u.cout))
        for up in self.ups:
            li = up.level_in
            if up.out is None:
                up.out = L.slack_empty((n, up.cout, 2 * hs[li], 2 * ws[li]), dev)
            if train and up.dout is None:
                up.dout = L.slack_empty((n, up.cout, 2 * hs[li], 2 * ws[li]), dev)   # ConvT dX reads pixel PAIRS: 2 floats past odd rows
            if up.wt_f is None or up.wt_f.device != dev:
                up.wt_f = torch.empty((lib.gsd_weight_layout_size(6, up.cout, up.cin),), **f32)
            if train:
                up.mode_d = lib.gsd_convT2x2_dgrad_layout(C.byref(L.make_src(up.dout, slack=L.SLACK)), up.cin, up.cout, n, hs[li], ws[li])
                need = lib.gsd_weight_layout_size(up.mode_d, up.cout, up.cin)
                if up.wt_d is None or up.wt_d.numel() != need or up.wt_d.device != dev:
                    up.wt_d = torch.empty((need,), **f32)
                # the LDS-DMA dX kernel can leave dz of the unit below and its per-channel sums (gsd_convT2x2_dgrad_bnrelu)
                up.bn_rows = 0
                if up.mode_d == 7 and self.convt_dg_bn:
                    up.bn_rows = lib.gsd_convT2x2_dgrad_bnrelu_partial_rows(C.byref(L.make_src(up.dout, slack=L.SLACK)), up.cin,
                                                                            up.cout, n, hs[li], ws[li])
                    max_part = max(max_part, up.bn_rows * 2 * _r64(up.cin))
            if train:
                max_ws = max(max_ws, lib.gsd_convT2x2_wgrad_workspace(n, hs[li], ws[li], up.cin, up.cout))
        self.pooled = [None] + [L.slack_empty((n, self.dims[l - 1], hs[l], ws[l]), dev) for l in range(1, self.L + 1)]
        self.dpooled = [None] + ([torch.empty((n, self.dims[l - 1], hs[l], ws[l]), **f32)
                                 for l in range(1, self.L + 1)] if train else [None] * self.L)
        # pitched d_raw scratch: one unit at a time on one stream; with dW on the side stream two, used in turn (dW of a unit may
        # still read its buffer while the BatchNorm backward of the next unit writes the other)
        side = train and self.side_dw
        self.side = torch.cuda.Stream(device=dev) if side else None   # (a high-priority side stream measured the same)
        self.gps = [torch.empty((max_gp,), **f32) for _ in range(2 if side else 1)] if (train and max_gp) else None
        self.gp_turn = 0
        self.gp_free: List[Optional[torch.cuda.Event]] = [None, None]   # recorded on the side stream behind a buffer's last reader
        self.partials = torch.empty((max_part,), **f32)
        self.conv_ws = torch.empty((max_slab,), **f32) if (train and max_slab) else None
        # fp64 column sums of a dX launch's statistics (ConvT bias gradients): gsd_bn_reduce_partials wants (1 + 64) x 2 C doubles
        self.db_sums = torch.empty((65 * 2 * max(u.cin for u in self.units),), device=dev, dtype=torch.float64) if train else None
        self.wgrad_ws = torch.empty((max(max_ws, 64 * max(1, self.n_classes)),), **f32) if train else None
        self.wgrad_ws_side = torch.empty((max(max_ws, 64),), **f32) if side else None
        self.outw_partials = self.outw_sums = None
        if train and self.n_classes > 1:    # dW of the output conv for K > 1 (gsd_conv1x1_out_wgrad)
            kc = self.n_classes * self.dims[0]
            self.outw_partials = torch.empty((lib.gsd_conv1x1_out_wgrad_rows(n, hs[0], ws[0]) * kc,), **f32)
            self.outw_sums = torch.empty((65 * kc,), device=dev, dtype=torch.float64)

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _layouts(jobs, st: int) -> None:
        """gsd_weight_layout_batch over (mode, weights, Co, Ci, image) tuples."""
        arr = (L.gsd_wl_job * len(jobs))()
        for i, (mode, w, co, ci, wt) in enumerate(jobs):
            arr[i].w, arr[i].wt, arr[i].mode, arr[i].Co, arr[i].Ci = w.data_ptr(), wt.data_ptr(), mode, co, ci
        check(lib.gsd_weight_layout_batch(arr, len(jobs), st), "weight_layout_batch")

    @staticmethod
    def _act_src(u: _Unit) -> L.gsd_src:
        return L.make_src(u.raw, u.scale, u.shift, relu=True, slack=L.SLACK)

    def _run_unit(self, u: _Unit, srcs: List[L.gsd_src], P: Dict[str, torch.Tensor], train: bool, st: int) -> None:
        n = u.raw.shape[0]
        lh, lw = self.hs[u.level], self.ws[u.level]
        u.form_f = u.forms_f[train]
        if not self.batch_wl:
            check(lib.gsd_weight_layout(u.form_f.mode_f, P[u.wname].data_ptr(), u.cout, u.cin, u.wt_f.data_ptr(), st), "weight_layout")
        arr = L.src_array(srcs)
        u.srcs = arr
        dst = L.dst_array([L.make_dst(u.raw)])
        part = self.partials.data_ptr() if train else None
        ev = self._log_begin()
        check(u.form_f.run(self.conv_ws if train else None, arr, len(srcs), u.wt_f.data_ptr(), u.cin, u.cout, dst, 1, part,
                           n, lh, lw, st), "conv3x3")
        self._log_end(ev, u.cout, u.cin, n, lh, lw, u.form_f.algo)
        if train:
            rows = u.form_f.partial_rows(n, lh, lw, u.cout)
            count = float(n * lh * lw)
            if self.sync_fn is None and rows <= self.one_launch_rows:
                # a few hundred partial rows (the deep levels; every level at small batches): column sums + finalize in ONE
                # launch instead of three (no SyncBN exchange in between)
                check(lib.gsd_bn_reduce_finalize(self.partials.data_ptr(), rows, _r64(u.cout), u.cout, u.sums.data_ptr(), count,
                                                 P[u.gname].data_ptr(), P[u.bname].data_ptr(), BN_EPS, BN_MOMENTUM,
                                                 P[u.rmname].data_ptr(), P[u.rvname].data_ptr(), u.mean.data_ptr(),
                                                 u.invstd.data_ptr(), u.scale.data_ptr(), u.shift.data_ptr(), self.guard, st),
                      "bn_reduce_finalize")
                self._nbt.append(P[u.nbtname])
                return
            check(lib.gsd_bn_reduce_partials(self.partials.data_ptr(), rows, _r64(u.cout), u.cout, u.sums.data_ptr(), st),
                  "bn_reduce_partials")
            if self.sync_fn is not None:
                self.sync_fn(u.sums[:2 * u.cout])
                count *= self.world
            check(lib.gsd_bn_finalize(u.sums.data_ptr(), u.cout, count, P[u.gname].data_ptr(), P[u.bname].data_ptr(),
                                      BN_EPS, BN_MOMENTUM, P[u.rmname].data_ptr(), P[u.rvname].data_ptr(),
                                      u.mean.data_ptr(), u.invstd.data_ptr(), u.scale.data_ptr(), u.shift.data_ptr(),
                                      self.guard, st),
                  "bn_finalize")
            self._nbt.append(P[u.nbtname])   # int64 counter buffers (BatchNorm2d.num_batches_tracked): one add for all, below
        else:
            check(lib.gsd_bn_eval_coeffs(P[u.gname].data_ptr(), P[u.bname].data_ptr(), P[u.rmname].data_ptr(),
                                         P[u.rvname].data_ptr(), BN_EPS, u.cout, u.scale.data_ptr(), u.shift.data_ptr(), st),
                  "bn_eval_coeffs")

    def _log_begin(self):
        if self.kernel_log is None and self.region_log is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _log_end(self, ev, m: int, k_ch: int, n: int, lh: int, lw: int, algo: int = 0) -> None:
        """(kernel, ALGORITHMIC flops of the convolution = 2*9*M*K*pixels, events, shape, flops the MFMAs executed)."""
        if ev is None or self.kernel_log is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        flops = 2.0 * m * k_ch * 9 * n * lh * lw
        if algo == 2:
            variant = "conv3x3_w2d_kernel"
            executed = 2048.0 * lib.gsd_conv3x3_w2d_mfma_count(n, lh, lw, k_ch, m)
        elif algo:
            variant = "conv3x3_w43_kernel"
            executed = 2048.0 * lib.gsd_conv3x3_w43_mfma_count(n, lh, lw, k_ch, m)   # one v_mfma_f32_16x16x4_f32 = 2048 flops
        else:
            variant = "conv3x3_dma_kernel<1,4>" if m <= 64 else "conv3x3_dma_kernel<2,2>"
            executed = flops
        self.kernel_log.append((variant, flops, ev, e, (m, k_ch, lh, lw), executed))

    def _pad_off(self, lvl: int) -> Tuple[int, int]:
        # F.pad(x1, [dX//2, dX-dX//2, dY//2, dY-dY//2]) (unet.py:43-47)
        dy = self.hs[lvl] - 2 * self.hs[lvl + 1]
        dx = self.ws[lvl] - 2 * self.ws[lvl + 1]
        return dy // 2, dx // 2

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, P: Dict[str, torch.Tensor], train: bool, out: Optional[torch.Tensor] = None
                ) -> torch.Tensor:
        """P: name -> tensor for every state_dict entry (reference names). Returns (N, n_classes, H, W)."""
        if x.dtype != torch.float32 or not x.is_cuda:
            raise L.GsdError("UNetEngine.forward needs a float32 tensor on the GPU (no CPU fallback)")
        self._nbt = []
        x = x.contiguous()
        n, c, h, w = x.shape
        assert c == self.n_channels, f"expected {self.n_channels} input channels, got {c}"
        self._ensure(n, h, w, x.device, train)
        st = L.stream_ptr()
        self._x = x
        self._saved_train = train
        self.generation += 1       # every forward overwrites the saved activations
        if self.batch_wl:          # every forward-mode weight layout of the pass: the 2-D Winograd images in one launch
            self._layouts([(u.forms_f[train].mode_f, P[u.wname], u.cout, u.cin, u.wt_f) for u in self.units] +
                          [(6, P[up.wname], up.cout, up.cin, up.wt_f) for up in self.ups], st)
        region = self._log_begin() if self.region_log is not None else None
        for lvl in range(self.L + 1):
            u0, u1 = self.enc[lvl]
            if lvl == 0:
                srcs = [L.make_src(x)]
            else:
                if lvl == 1 and region is not None:       # bench hook: the `inc` double-conv forward (2 convs + BN statistics)
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    self.region_log.append(("inc_forward", region, e))
                prev = self.enc[lvl - 1][1]
                s = self._act_src(prev)
                check(lib.gsd_maxpool2(C.byref(s), self.pooled[lvl].data_ptr(), n, prev.cout, self.hs[lvl - 1],
                                       self.ws[lvl - 1], st), "maxpool2")
                srcs = [L.make_src(self.pooled[lvl], slack=L.SLACK)]
            self._run_unit(u0, srcs, P, train, st)
            self._run_unit(u1, [self._act_src(u0)], P, train, st)
        if self.L == 0 and region is not None:          # a one-level network (profiles/inc_block.py): the block ends here
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.region_log.append(("inc_forward", region, e))
        cur = self.enc[self.L][1]
        for j in range(self.L):
            up = self.ups[j]
            lvl = self.L - 1 - j
            if not self.batch_wl:
                check(lib.gsd_weight_layout(6, P[up.wname].data_ptr(), up.cout, up.cin, up.wt_f.data_ptr(), st), "weight_layout")
            s = self._act_src(cur)
            d = L.make_dst(up.out)
            check(lib.gsd_convT2x2(C.byref(s), up.wt_f.data_ptr(), P[up.bname].data_ptr(), up.cin, up.cout, C.byref(d), n,
                                   self.hs[lvl + 1], self.ws[lvl + 1], st), "convT2x2")
            skip = self.enc[lvl][1]
            u0, u1 = self.dec[j]
            self._run_unit(u0, [self._act_src(skip), L.make_src(up.out, off=self._pad_off(lvl), slack=L.SLACK)], P, train, st)
            self._run_unit(u1, [self._act_src(u0)], P, train, st)
            cur = u1
        if self._nbt:
            L.add_counters(self._nbt, 1)        # one libgsd launch for every BatchNorm layer's counter
            self._nbt = []
        if out is None:
            out = torch.empty((n, self.n_classes, h, w), device=x.device, dtype=torch.float32)
        s = self._act_src(cur)
        check(lib.gsd_conv1x1_out(C.byref(s), P["outc.conv.weight"].data_ptr(), P["outc.conv.bias"].data_ptr(), cur.cout,
                                  self.n_classes, out.data_ptr(), n, h, w, st), "conv1x1_out")
        return out

    # ------------------------------------------------------------------ backward
    def _bn_bwd_tail(self, u: _Unit, P, G, st: int, dwout: Optional[torch.Tensor] = None, fused: bool = False) -> None:
        """u.g holds dz and self.partials its per-block sums: finish BN backward, then dW.
        fused: the partials come from gsd_conv3x3_dgrad_bnrelu (conv layout) instead of gsd_bn_bwd_reduce."""
        n = u.raw.shape[0]
        lh, lw = self.hs[u.level], self.ws[u.level]
        rows = u.fused_rows if fused else lib.gsd_bn_bwd_partial_rows(n, u.cout, lh, lw)   # fused: of the dX launch (_dgrad_fused)
        count = float(n * lh * lw)
        if self.sync_fn is None and rows <= self.one_launch_rows and (dwout is None or not fused):
            check(lib.gsd_bn_bwd_reduce_finalize(self.partials.data_ptr(), rows, _r64(u.cout) if fused else 0, u.cout,
                                                 u.sums.data_ptr(), count, G[u.gname].data_ptr(), G[u.bname].data_ptr(),
                                                 None if dwout is None else dwout.data_ptr(), u.c1.data_ptr(), u.c2.data_ptr(), st),
                  "bn_bwd_reduce_finalize")
        else:
            if fused:
                check(lib.gsd_bn_reduce_partials(self.partials.data_ptr(), rows, _r64(u.cout), u.cout, u.sums.data_ptr(), st),
                      "bn_reduce_partials")
            else:
                check(lib.gsd_bn_bwd_reduce_partials(self.partials.data_ptr(), rows, u.cout, u.sums.data_ptr(), st),
                      "bn_bwd_reduce_partials")
            gsum = None
            if self.sync_fn is not None:
                gsum = u.sums[:2 * u.cout].clone()
                self.sync_fn(gsum)
                count *= self.world
            check(lib.gsd_bn_bwd_finalize(u.sums.data_ptr(), L.ptr(gsum), u.cout, count, G[u.gname].data_ptr(),
                                          G[u.bname].data_ptr(), None if dwout is None else dwout.data_ptr(),
                                          u.c1.data_ptr(), u.c2.data_ptr(), st),
                  "bn_bwd_finalize")
        if u.fused_dw:
            self._on_side(lambda sst, ws: check(
                lib.gsd_conv3x3_wgrad_bn(u.srcs, u.g.data_ptr(), u.raw.data_ptr(), u.scale.data_ptr(), u.mean.data_ptr(),
                                         u.invstd.data_ptr(), u.c1.data_ptr(), u.c2.data_ptr(), u.cin, u.cout,
                                         G[u.wname].data_ptr(), ws.data_ptr(), ws.numel(), n, lh, lw, sst), "conv3x3_wgrad_bn"))
            return
        turn = None
        if u.pitched:
            p = _r4(lw)
            turn = self.gp_turn = (self.gp_turn + 1) % len(self.gps)
            if self.gp_free[turn] is not None:     # the dW launch that read this buffer two units ago
                torch.cuda.current_stream().wait_event(self.gp_free[turn])
                self.gp_free[turn] = None
            u.dsrc = self.gps[turn][:n * u.cout * lh * p].view(n, u.cout, lh, p)[..., :lw]
            out_ptr = u.dsrc.data_ptr()
        else:
            p, u.dsrc, out_ptr = 0, u.g, None
        check(lib.gsd_bn_bwd_apply(u.g.data_ptr(), u.raw.data_ptr(), u.scale.data_ptr(), u.mean.data_ptr(),
                                   u.invstd.data_ptr(), u.c1.data_ptr(), u.c2.data_ptr(), n, u.cout, lh, lw, out_ptr, p, st),
              "bn_bwd_apply")
        dy = L.make_src(u.dsrc)
        ev0 = self._log_begin() if self.wgrad_log is not None else None
        on_side = self._on_side(lambda sst, ws: check(
            lib.gsd_conv3x3_wgrad(u.srcs, len(u.srcs), C.byref(dy), u.cin, u.cout, G[u.wname].data_ptr(), ws.data_ptr(), ws.numel(),
                                  n, lh, lw, sst), "conv3x3_wgrad"))
        if ev0 is not None and not on_side:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            form = lib.gsd_conv3x3_wgrad_form(u.srcs, len(u.srcs), C.byref(dy), u.cin, u.cout, n, lh, lw)
            name = ("wgrad3x3_kernel", "wgrad3x3_w43_kernel", "wgrad3x3_w2d_kernel")[form]
            flops = 2.0 * u.cout * u.cin * 9 * n * lh * lw
            executed = 2048.0 * lib.gsd_conv3x3_wgrad_mfma_count(form, n, lh, lw, u.cin, u.cout) if form else flops
            self.wgrad_log.append((name, flops, ev0, ev1, (u.cout, u.cin, lh, lw), executed))
        if on_side and turn is not None:
            self.gp_free[turn] = self.side.record_event()

    def _on_side(self, launch) -> bool:
        """Run launch(stream pointer, workspace tensor) -- one weight-gradient launch -- behind everything issued so far: on the
        side stream when there is one (True), else on the current stream."""
        if self.side is None or self.kernel_log is not None:    # (a per-kernel timing pass wants every kernel alone on the chip)
            launch(L.stream_ptr(), self.wgrad_ws)
            return False
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            launch(L.stream_ptr(), self.wgrad_ws_side)
        return True

    def _join_side(self) -> None:
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
            self.gp_free = [None, None]

    def _announce(self, tag: str) -> None:
        """A block of gradients is final once BOTH streams have finished what was issued so far.  The callback (the bucket's
        all-reduce: RCCL orders itself behind the stream that is current when it is called) therefore runs under a hand-off
        stream that waits for the two -- the main stream does NOT: the dX chain of the next unit does not depend on this
        block's weight gradients and used to stall behind them nine times per step."""
        if self.block_done_cb is None:
            return
        if self.side is None or self.kernel_log is not None:
            self.block_done_cb(tag)
            return
        if self._handoff is None:
            self._handoff = torch.cuda.Stream(device=self.side.device)
        h = self._handoff
        h.wait_stream(torch.cuda.current_stream())
        h.wait_stream(self.side)
        with torch.cuda.stream(h):
            self.block_done_cb(tag)

    def _reduce(self, mode: int, u: _Unit, st: int, dpool: Optional[torch.Tensor] = None,
                dout: Optional[torch.Tensor] = None, wout: Optional[torch.Tensor] = None) -> None:
        n = u.raw.shape[0]
        lh, lw = self.hs[u.level], self.ws[u.level]
        da = L.make_src(u.g)
        check(lib.gsd_bn_bwd_reduce(mode, u.raw.data_ptr(), u.scale.data_ptr(), u.shift.data_ptr(), u.mean.data_ptr(),
                                    u.invstd.data_ptr(), C.byref(da), L.ptr(dpool), L.ptr(dout), L.ptr(wout),
                                    self.n_classes, u.g.data_ptr(), self.partials.data_ptr(), n, u.cout, lh, lw, st),
              "bn_bwd_reduce")

    def _dgrad(self, u: _Unit, P, dsts: List[L.gsd_dst], st: int, stats: bool = False) -> int:
        """dX of unit u into the destination segments.  stats: the launch also leaves per-channel partial sums of what it
        stored in self.partials (the conv epilogue's BatchNorm-statistics path); returns their row count."""
        n = u.raw.shape[0]
        lh, lw = self.hs[u.level], self.ws[u.level]
        if not self.batch_wl:
            check(lib.gsd_weight_layout(u.form_d.mode_d, P[u.wname].data_ptr(), u.cout, u.cin, u.wt_d.data_ptr(), st), "weight_layout")
        s = L.src_array([L.make_src(u.dsrc)])
        ev = self._log_begin()
        check(u.form_d.run(self.conv_ws, s, 1, u.wt_d.data_ptr(), u.cout, u.cin, L.dst_array(dsts), len(dsts),
                           self.partials.data_ptr() if stats else None, n, lh, lw, st), "conv3x3 dgrad")
        self._log_end(ev, u.cin, u.cout, n, lh, lw, u.form_d.algo)
        return u.form_d.partial_rows(n, lh, lw, u.cin) if stats else 0

    def _dgrad_fused(self, u: _Unit, prev: _Unit, P, st: int) -> None:
        """dX of unit u straight into prev.g as dz of prev's relu(bn(.)) (+ partial sums): u's input is prev's output."""
        n = u.raw.shape[0]
        lh, lw = self.hs[u.level], self.ws[u.level]
        if not self.batch_wl:
            check(lib.gsd_weight_layout(u.form_d.mode_d, P[u.wname].data_ptr(), u.cout, u.cin, u.wt_d.data_ptr(), st), "weight_layout")
        s = L.make_src(u.dsrc)
        d = L.make_dst(prev.g)
        prev.fused_rows = u.form_d.partial_rows(n, lh, lw, u.cin)
        ev = self._log_begin()
        check(u.form_d.run_bnrelu(self.conv_ws, C.byref(s), u.wt_d.data_ptr(), u.cout, u.cin, C.byref(d), prev.raw.data_ptr(),
                                  prev.scale.data_ptr(), prev.shift.data_ptr(), prev.mean.data_ptr(),
                                  prev.invstd.data_ptr(), self.partials.data_ptr(), n, lh, lw, st),
              "conv3x3_dgrad_bnrelu")
        self._log_end(ev, u.cin, u.cout, n, lh, lw, u.form_d.algo)

    def backward(self, dout: torch.Tensor, P: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor]) -> None:
        """dout: (N, n_classes, H, W) gradient of the loss w.r.t. the output.
        G: name -> tensor to receive every parameter's gradient (overwritten, not accumulated)."""
        if not self._saved_train:
            raise L.GsdError("backward() needs a preceding train-mode forward()")
        dout = dout.contiguous()
        st = L.stream_ptr()
        n = dout.shape[0]
        if self.batch_wl:          # the dX-mode weight layouts (the weights have not changed since the forward)
            self._layouts([(u.form_d.mode_d, P[u.wname], u.cout, u.cin, u.wt_d) for u in self.units if u.need_dgrad] +
                          [(up.mode_d, P[up.wname], up.cout, up.cin, up.wt_d) for up in self.ups], st)
        last = self.dec[-1][1] if self.L > 0 else self.enc[0][1]
        self._reduce(2, last, st, dout=dout, wout=P["outc.conv.weight"])
        check(lib.gsd_sum_planes(dout.data_ptr(), n, self.n_classes, dout.shape[2] * dout.shape[3],
                                 G["outc.conv.bias"].data_ptr(), self.wgrad_ws.data_ptr(), st), "sum_planes")
        dwout = G["outc.conv.weight"]
        if self.n_classes > 1:
            # K > 1: the reduce above leaves only row 0 of dW_out among its sums; all K rows come from a pass of their own
            # (aten::convolution_backward of unet.py:54; no reference config uses it, so it is not on the tuned path)
            lh, lw = self.hs[last.level], self.ws[last.level]
            check(lib.gsd_conv1x1_out_wgrad(last.raw.data_ptr(), last.scale.data_ptr(), last.shift.data_ptr(), dout.data_ptr(),
                                            last.cout, self.n_classes, dwout.data_ptr(), self.outw_partials.data_ptr(),
                                            self.outw_sums.data_ptr(), n, lh, lw, st), "conv1x1_out_wgrad")
            dwout = None
        for j in reversed(range(self.L)):
            u0, u1 = self.dec[j]
            up = self.ups[j]
            lvl = self.L - 1 - j
            # dz of u1 came from the output conv (j = L-1) or from the ConvT dX of the level above -- with its sums when fused
            self._bn_bwd_tail(u1, P, G, st, dwout, fused=j < self.L - 1 and self.ups[j + 1].bn_rows > 0)
            dwout = None
            self._dgrad_fused(u1, u0, P, st)
            self._bn_bwd_tail(u0, P, G, st, fused=True)
            skip = self.enc[lvl][1]
            # the ConvT bias gradient is the per-channel sum of up.dout: the Winograd dX launch that writes up.dout leaves it
            # as statistics of its second (cropped) destination -- no second pass over up.dout
            db_fused = u0.form_d.algo >= 1
            rows = self._dgrad(u0, P, [L.make_dst(skip.g), L.make_dst(up.dout, off=self._pad_off(lvl))], st, stats=db_fused)
            if db_fused:
                check(lib.gsd_partials_channel_sums(self.partials.data_ptr(), rows, _r64(u0.cin), u0.cin, skip.cout, up.cout,
                                                    G[up.bname].data_ptr(), self.db_sums.data_ptr(), st), "partials_channel_sums")
            prev = self.dec[j - 1][1] if j > 0 else self.enc[self.L][1]
            hi, wi = self.hs[lvl + 1], self.ws[lvl + 1]
            xs = self._act_src(prev)
            dys = L.make_src(up.dout, slack=L.SLACK)
            self._on_side(lambda sst, ws, up=up, xs=xs, dys=dys, db_fused=db_fused, hi=hi, wi=wi: check(
                lib.gsd_convT2x2_wgrad(C.byref(xs), C.byref(dys), up.cin, up.cout, G[up.wname].data_ptr(),
                                       None if db_fused else G[up.bname].data_ptr(), ws.data_ptr(), ws.numel(), n, hi, wi, sst),
                "convT2x2_wgrad"))
            if not self.batch_wl:
                check(lib.gsd_weight_layout(up.mode_d, P[up.wname].data_ptr(), up.cout, up.cin, up.wt_d.data_ptr(), st), "weight_layout")
            d = L.make_dst(prev.g)
            if up.bn_rows:
                check(lib.gsd_convT2x2_dgrad_bnrelu(C.byref(dys), up.wt_d.data_ptr(), up.cin, up.cout, C.byref(d), prev.raw.data_ptr(),
                                                    prev.scale.data_ptr(), prev.shift.data_ptr(), prev.mean.data_ptr(),
                                                    prev.invstd.data_ptr(), self.partials.data_ptr(), n, hi, wi, st),
                      "convT2x2_dgrad_bnrelu")
                prev.fused_rows = up.bn_rows
            else:
                check(lib.gsd_convT2x2_dgrad_as(up.mode_d, C.byref(dys), up.wt_d.data_ptr(), up.cin, up.cout, C.byref(d), n, hi, wi, st),
                      "convT2x2_dgrad")
                self._reduce(0, prev, st)
            self._announce(f"dec{j}")      # up.{j}.* (and outc with the last decoder) are final
        for lvl in reversed(range(self.L + 1)):
            u0, u1 = self.enc[lvl]
            if lvl < self.L:
                self._reduce(1, u1, st, dpool=self.dpooled[lvl + 1])
            # the bottom unit's dz came from the first ConvT dX -- with its sums when that launch was the fused one
            self._bn_bwd_tail(u1, P, G, st, dwout, fused=lvl == self.L and self.L > 0 and self.ups[0].bn_rows > 0)
            dwout = None
            self._dgrad_fused(u1, u0, P, st)
            self._bn_bwd_tail(u0, P, G, st, fused=True)
            self._announce(f"enc{lvl}")
            if lvl > 0:
                self._dgrad(u0, P, [L.make_dst(self.dpooled[lvl])], st)
        self._join_side()
