"""The reference's train-step body as one fused schedule of libgsd kernels.

Reference (paths under /root/reference/):
    optimizer.zero_grad(); output = unet(x=input_image)            train_utils/train_unet.py:346-347
    pred_loss = MSE_loss(input=output, target=output_target)       :370  (def :51-52)
    loss.backward(); optimizer.step(); ema.update()                :374-376
    Adam(lr=1e-3, weight_decay=1e-6) :306, ExponentialMovingAverage(decay=0.995) :309

Here: forward -> loss+grad kernel -> backward -> (RCCL all-reduce of the flat gradient arena, bucketed and
overlapped with the rest of backward) -> one fused Adam+EMA kernel over the flat parameter arena.
No autograd graph, no per-tensor optimiser launches, no host sync inside the step (the loss stays on the
device; call .item() when you want it).

The reference's NaN guard (train_unet.py:371-372) replaces a NaN loss by a constant without grad_fn, after which
loss.backward() raises; it costs a host sync per step.  Here it is a device-side flag (`nan_policy`, gsd_guard in
include/gsd.h): a step whose loss or BatchNorm batch statistics are non-finite leaves parameters, Adam moments and the
EMA shadow untouched and is counted on the device (non-finite batch statistics never reach the running statistics, with
or without a policy).  A skipped step still consumes a tick of the two host-side schedules -- Adam's bias-correction step
count and torch_ema's warm-up count ((1+n)/(10+n)) advance as if the step had been applied, where torch / torch_ema would
not have counted it; either factor changes by less than 1/t from tick t to t+1, so the effect of a skipped step fades with
the step count -- "skip" carries on, "raise" raises GsdError at
the next `check_finite()` (train_epoch calls it once per epoch: one host sync per epoch instead of two per step),
None (default) runs the reference's arithmetic unguarded.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from ._lib import lib, check
from .models.unet import UNet

LOSS_KINDS = {"mse": 0, "l1": 1}


def loss_fwd_bwd(kind: str, out: torch.Tensor, target: torch.Tensor, grad: Optional[torch.Tensor],
                 loss_buf: torch.Tensor, ws: torch.Tensor, grad_scale: float = 1.0, guard=None) -> None:
    """loss_buf[0] = mean((o-t)^2) | mean(|o-t|); grad = d loss / d out * grad_scale (train_unet.py:51-52)."""
    for name, t in (("output", out), ("target", target)):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise L.GsdError(f"loss_fwd_bwd: {name} must be a contiguous float32 tensor on the GPU, got {t.dtype} on "
                             f"{t.device} (the kernel reads raw fp32; cast with .float() first)")
    if out.shape != target.shape:
        raise L.GsdError(f"loss_fwd_bwd: output {tuple(out.shape)} and target {tuple(target.shape)} differ in shape")
    check(lib.gsd_loss_fwd_bwd(LOSS_KINDS[kind], out.data_ptr(), target.data_ptr(), out.numel(), grad_scale,
                               loss_buf.data_ptr(), L.ptr(grad), ws.data_ptr(), guard, L.stream_ptr()), "loss_fwd_bwd")


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, target, kind):
        out_c, tgt_c = out.contiguous(), target.contiguous().float()
        loss = torch.empty((1,), device=out.device, dtype=torch.float32)
        grad = torch.empty_like(out_c)
        ws = torch.empty((2048,), device=out.device, dtype=torch.float64)
        loss_fwd_bwd(kind, out_c, tgt_c, grad, loss, ws)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def mse_loss(input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Drop-in for the reference's MSE_loss(input, target) (train_unet.py:51-52), libgsd kernel."""
    return _LossFn.apply(input, target, "mse")


def l1_loss(input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return _LossFn.apply(input, target, "l1")


class TrainStep:
    """Fused fwd + loss + bwd + Adam + EMA step for a gelslim_depth_amd UNet.

    Parameters live in one flat fp32 arena (the module's parameters become views of it, so state_dict(),
    load_state_dict() and the reference's checkpoint layout keep working); gradients, Adam moments and the EMA
    shadow are arenas of the same size.  With `process_group` set, gradients are summed across ranks with
    RCCL all-reduce (torch.distributed backend "nccl") in per-block buckets launched as soon as a block's
    backward is done, and scaled by 1/world inside the Adam kernel.
    """

    def __init__(self, model: UNet, lr: float = 1e-3, weight_decay: float = 1e-6, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, ema_decay: Optional[float] = 0.995, loss: str = "mse",
                 process_group=None, sync_bn: bool = False, overlap_allreduce: bool = True,
                 nan_policy: Optional[str] = None, force_sync: bool = False, time_allreduce: bool = False):
        if nan_policy not in (None, "skip", "raise"):
            raise ValueError(f"nan_policy must be None, 'skip' or 'raise', got {nan_policy!r}")
        self.model = model
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.ema_decay = ema_decay
        self.loss_kind = loss
        self.step_count = 0
        self.ema_updates = 0
        self.pg = process_group
        self.world, self.rank = 1, 0
        self.overlap = overlap_allreduce
        self.nan_policy = nan_policy
        if process_group is not None:
            import torch.distributed as dist
            self.dist = dist
            self.world = dist.get_world_size(process_group)
            self.rank = dist.get_rank(process_group)
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise L.GsdError("TrainStep needs the model on the GPU")
        names = [n for n, _ in model.named_parameters()]
        sizes = [p.numel() for _, p in model.named_parameters()]
        total = sum(sizes)
        self.numel = total
        self.p_flat = torch.empty((total,), device=dev, dtype=torch.float32)
        self.g_flat = torch.zeros((total,), device=dev, dtype=torch.float32)
        self.m_flat = torch.zeros((total,), device=dev, dtype=torch.float32)
        self.v_flat = torch.zeros((total,), device=dev, dtype=torch.float32)
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        gviews: Dict[str, torch.Tensor] = {}
        for (n, p), sz in zip(model.named_parameters(), sizes):
            self.p_flat[off:off + sz].copy_(p.data.reshape(-1))
            p.data = self.p_flat[off:off + sz].view(p.shape)
            gviews[n] = self.g_flat[off:off + sz].view(p.shape)
            self.offsets[n] = (off, sz)
            off += sz
        model._grad_views = gviews
        self.ema_flat = self.p_flat.clone() if ema_decay is not None else None
        self.loss_buf = torch.zeros((1,), device=dev, dtype=torch.float32)
        self.loss_ws = torch.empty((2048,), device=dev, dtype=torch.float64)
        # non-finite guard: words[0] = tick of the last bad step, words[1] = steps skipped (include/gsd.h: gsd_guard)
        self.guard_words = torch.zeros((2,), device=dev, dtype=torch.int32) if nan_policy is not None else None
        # With a nan_policy a skipped step must leave no trace in the BatchNorm running statistics either (layers in front of
        # the first bad one have updated theirs by the time the step is found bad, and every data-parallel rank must end with
        # the same buffers): the float buffers become views of one arena that is snapshotted before the step and put back
        # by a device-side conditional copy behind it.  (num_batches_tracked keeps counting, as it would in the reference,
        # whose forward has run by the time its NaN test fires.)
        self.bn_flat = self.bn_snap = None
        if nan_policy is not None:
            bufs = [b for _, b in model.named_buffers() if b.dtype == torch.float32]
            self.bn_flat = torch.empty((sum(b.numel() for b in bufs),), device=dev, dtype=torch.float32)
            o = 0
            for b in bufs:
                self.bn_flat[o:o + b.numel()].copy_(b.reshape(-1))
                b.data = self.bn_flat[o:o + b.numel()].view(b.shape)
                o += b.numel()
            self.bn_snap = torch.empty_like(self.bn_flat)
        # A later model.to(...) / .float() / .half() re-allocates the module's tensors and silently detaches them from the arenas
        # (the kernels would go on updating arenas nobody reads; the guard's snapshot would cover stale memory): remember where
        # every parameter and arena-backed buffer must live and check it at every step (a host-side pointer compare).
        self._pinned = [(n, p, p.data_ptr()) for n, p in model.named_parameters()]
        if self.bn_flat is not None:
            self._pinned += [(n, b, b.data_ptr()) for n, b in model.named_buffers() if b.dtype == torch.float32]
        self._dout = None
        self._out = None
        eng = model._engine
        eng.world = self.world
        force_sync = process_group is not None and (force_sync or bool(os.environ.get("GSD_FORCE_SYNC")))
        if sync_bn and (self.world > 1 or force_sync):
            eng.sync_fn = lambda t: self.dist.all_reduce(t, group=self.pg)
        self.sync = None
        if self.world > 1 or force_sync:   # 2nd: rehearsal of the collectives with ONE rank
            from .distributed import GradSync, broadcast_state, make_buckets
            broadcast_state(self.p_flat, [b for _, b in model.named_buffers()], group=self.pg)
            if self.ema_flat is not None:
                self.ema_flat.copy_(self.p_flat)
            self.sync = GradSync(self.g_flat, make_buckets(names, self.offsets, eng.L), group=self.pg,
                                 overlap=overlap_allreduce, force=force_sync, timing=time_allreduce)

    def _check_arenas(self) -> None:
        for n, t, ptr in self._pinned:
            if t.data_ptr() != ptr or t.dtype != torch.float32:
                raise L.GsdError(
                    f"TrainStep: '{n}' no longer lives in the step's flat arena (the model was moved or cast -- model.to(...), "
                    ".float(), .half(), load with assign=True -- after TrainStep was built).  Move / cast the model first, then "
                    "construct TrainStep; load_state_dict() and in-place updates keep the arenas.")

    def __call__(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        model = self.model
        eng = model._engine
        self._check_arenas()
        x = x.contiguous()
        target = target.float().contiguous()     # the kernels read raw fp32 (a float64 / uint8 depth target is cast, as _LossFn does)
        if not target.is_cuda:
            raise L.GsdError("TrainStep: the target must be on the GPU")
        guard = L.make_guard(self.guard_words, self.step_count + 1)
        eng.guard = guard
        P = model._tensor_map()
        if self._out is None or self._out.shape != (x.shape[0], model.n_classes, x.shape[2], x.shape[3]):
            self._out = torch.empty((x.shape[0], model.n_classes, x.shape[2], x.shape[3]), device=x.device,
                                    dtype=torch.float32)
            self._dout = torch.empty_like(self._out)
        if self.bn_flat is not None:
            check(lib.gsd_guard_snapshot(self.bn_flat.data_ptr(), self.bn_snap.data_ptr(), self.bn_flat.numel(), L.stream_ptr()),
                  "guard_snapshot")
        try:
            out = eng.forward(x, P, train=True, out=self._out)
            loss_fwd_bwd(self.loss_kind, out, target, self._dout, self.loss_buf, self.loss_ws, guard=guard)
            eng.block_done_cb = self.sync.on_block_done if self.sync is not None else None
            eng.backward(self._dout, P, model._grad_views)
        finally:
            eng.block_done_cb = None
            eng.guard = None
        if self.sync is not None:
            self.sync.finish()
            if guard is not None:     # every rank must take the same skip decision: the summed gradient carries any rank's NaN
                self.dist.all_reduce(self.guard_words[0:1], op=self.dist.ReduceOp.MAX, group=self.pg)
        self.step_count += 1
        d = 0.0
        if self.ema_flat is not None:
            # torch_ema 0.3 (requirements.txt:6): decay = min(decay, (1+n)/(10+n)), n counted after increment
            self.ema_updates += 1
            d = min(self.ema_decay, (1.0 + self.ema_updates) / (10.0 + self.ema_updates))
        check(lib.gsd_adam_ema(self.p_flat.data_ptr(), self.g_flat.data_ptr(), self.m_flat.data_ptr(),
                               self.v_flat.data_ptr(), L.ptr(self.ema_flat), self.numel, self.step_count, self.lr,
                               self.betas[0], self.betas[1], self.eps, self.wd, d, 1.0 / self.world, guard, L.stream_ptr()),
              "adam_ema")
        if self.bn_flat is not None:
            check(lib.gsd_guard_restore(guard, self.bn_flat.data_ptr(), self.bn_snap.data_ptr(), self.bn_flat.numel(), L.stream_ptr()),
                  "guard_restore")
        return self.loss_buf

    def skipped_steps(self) -> int:
        """Optimiser steps the non-finite guard has skipped so far (one host sync; 0 without a nan_policy)."""
        return int(self.guard_words[1].item()) if self.guard_words is not None else 0

    def check_finite(self) -> None:
        """nan_policy="raise": raise if any step since the last check saw a non-finite loss or BatchNorm statistic (the
        reference fails at that step, train_unet.py:371-374; here the arenas were left untouched by the skipped update)."""
        if self.nan_policy != "raise":
            return
        n = self.skipped_steps()
        if n:
            bad = int(self.guard_words[0].item())
            self.guard_words[1].zero_()
            raise L.GsdError(f"non-finite loss or BatchNorm statistics in {n} train step(s), last at step {bad}; those "
                             "updates were skipped (parameters, Adam moments and EMA shadow are intact)")

    def mean_across_ranks(self, value: float) -> float:
        """Mean of a host scalar over the data-parallel ranks (epoch losses: every rank must take the same early-stopping
        and checkpoint decisions)."""
        if self.world == 1:
            return float(value)
        t = torch.tensor([value], device=self.p_flat.device, dtype=torch.float64)
        self.dist.all_reduce(t, group=self.pg)
        return float(t.item()) / self.world

    def evaluate(self, x: torch.Tensor, use_ema: bool = True) -> torch.Tensor:
        """Eval-mode forward under the EMA weights WITHOUT the store / copy-in / restore the reference pays per batch
        (`with ema.average_parameters(): unet(x=...)`, train_unet.py:389-390,428-429): the kernels
        are simply pointed at the shadow arena.  BatchNorm buffers are the live ones, as in the reference."""
        model = self.model
        P = model._tensor_map()
        if use_ema and self.ema_flat is not None:
            for k, (o, sz) in self.offsets.items():
                P[k] = self.ema_flat[o:o + sz].view(P[k].shape)
        with torch.no_grad():
            return model._engine.forward(x.float().contiguous(), P, train=False)

    def save_checkpoint(self, path: str, use_ema: bool = True) -> None:
        """torch.save of the reference-layout state_dict (118 keys), EMA weights swapped in like the reference's
        best-validation save (train_unet.py:480-483); loads into the reference's UNet with strict=True."""
        sd = self.ema_state_dict() if use_ema else self.model.state_dict()
        torch.save({k: v.detach().cpu() for k, v in sd.items()}, path)

    @property
    def last_loss(self) -> torch.Tensor:
        return self.loss_buf

    def ema_state_dict(self) -> Dict[str, torch.Tensor]:
        """state_dict with the EMA shadow in place of the parameters -- what the reference saves at best-val
        under `with ema.average_parameters()` (train_unet.py:480-483); BN buffers are the live ones."""
        sd = self.model.state_dict()
        if self.ema_flat is None:
            return sd
        out = {}
        for k, v in sd.items():
            if k in self.offsets:
                o, s = self.offsets[k]
                out[k] = self.ema_flat[o:o + s].view(v.shape).clone()
            else:
                out[k] = v.clone()
        return out
