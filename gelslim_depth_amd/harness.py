"""Epoch loop around the fused step: the reference's training harness without its plotting.

Mirrors /root/reference/train_utils/train_unet.py:312-523:
  * per epoch: train pass (:337-378), validation and test passes in eval mode under the EMA weights with NaN losses
    counted as 0 (:379-458), each loss averaged over the loader's batches;
  * validation-loss smoothing: a ring of the last `val_loss_SMA_window` validation losses, initialised to ZEROS, whose mean
    is compared with the previous epoch's mean; more than `validation_loss_count_threshold` consecutive rises stop the
    run -- or, with `train_indefinitely`, are only logged (:460-474);
  * checkpoint of the EMA-swapped, reference-layout state_dict whenever the raw validation loss reaches a new minimum
    (:475-483), and `_epoch{e}` snapshots at `save_at_epochs` (0-based epoch index, :484-489);
  * the text log: the same lines, in the same order and format, as the reference writes to its loss file (:465,478,
    490-496,520-523).

The device work is libgsd's: TrainStep for the train pass, TrainStep.evaluate (kernels pointed at the EMA arena, no
store/copy/restore) + the fused loss kernel for the other two.  `max_epochs` is an addition: the reference with
train_indefinitely=True only stops when killed.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np


class EarlyStopping:
    """The reference's stopping rule as a small state machine (train_unet.py:316-323, 460-474)."""

    def __init__(self, window: int = 10, count_threshold: int = 5, train_indefinitely: bool = False) -> None:
        self.window, self.count_threshold, self.train_indefinitely = window, count_threshold, train_indefinitely
        self.validation_array = np.zeros(window)
        self.prev_validation_loss = 0.0
        self.validation_loss_upward_counter = 0
        self.min_validation_loss = 1000000
        self.e = 0

    def update(self, validation_loss: float):
        """Feed epoch e's validation loss.  Returns (stop, stalled_message_needed, new_minimum)."""
        self.validation_array[self.e % self.window] = validation_loss
        smoothed = float(np.mean(self.validation_array))
        if smoothed > self.prev_validation_loss:
            self.validation_loss_upward_counter += 1
        else:
            self.validation_loss_upward_counter = 0
        stop = stalled = False
        if self.validation_loss_upward_counter > self.count_threshold:
            stop = True
            if self.train_indefinitely:
                stalled, stop = True, False
        self.prev_validation_loss = smoothed
        new_min = validation_loss < self.min_validation_loss
        if new_min:
            self.min_validation_loss = validation_loss
        self.e += 1
        return stop, stalled, new_min


def _mean_loss(losses: List[float], n_batches: int) -> float:
    return float(sum(losses) / n_batches) if n_batches else 0.0


def evaluate_loader(step, loader: Iterable[Dict], loss_kind: str = "mse") -> float:
    """Mean over the loader's batches of the loss under the EMA weights, NaN batches counted as 0 (train_unet.py:379-419).

    Data parallel (a `DeviceLoader` with world_size > 1): the walk is over the GLOBAL batches, every rank evaluates only its
    contiguous share of each at the per-rank train shape (`DeviceLoader.eval_shares`: no wrap-around padding, no activation
    buffer reallocated, nothing evaluated twice), and ONE all-reduce at the end of the pass sums (loss sum, element count) per
    global batch -- every rank then holds the single-process value of every batch loss, bit for bit the same on all ranks.
    The batch losses stay on the device until the pass ends (one host sync per pass; the reference syncs per batch)."""
    import torch
    from .train import loss_fwd_bwd
    sharded = getattr(loader, "world_size", 1) > 1 and hasattr(loader, "eval_shares")
    if sharded and (getattr(step, "pg", None) is None or getattr(step, "dist", None) is None):
        raise RuntimeError("evaluate_loader: the loader is sharded over %d ranks but the step was built without a process_group; "
                           "pass the group to TrainStep or evaluate loader.unsharded()" % loader.world_size)
    buf = ws = None
    rows = []          # per (global) batch: device tensor [loss sum over this rank's valid elements, their count]
    dev = None
    walk = loader.eval_shares() if sharded else ((data, None, None) for data in loader)
    for data, valid, _ in walk:
        if data is None:                  # the ragged tail left this rank nothing of this global batch
            rows.append(None)
            continue
        x, t = data["tactile_image"], data["depth_image"]
        out = step.evaluate(x, use_ema=True)
        if buf is None:
            dev = out.device
            buf = torch.zeros((1,), device=dev, dtype=torch.float32)
            ws = torch.empty((2048,), device=dev, dtype=torch.float64)
        t = t.float().contiguous()
        if valid is not None and valid < out.shape[0]:
            out, t = out[:valid], t[:valid]       # leading-dimension slices stay contiguous: the padding is not scored
        loss_fwd_bwd(loss_kind, out, t, None, buf, ws)
        rows.append((buf[0].double().clone(), float(out.numel())))      # the element counts stay on the host
    if not rows:
        return 0.0
    if dev is None:
        dev = getattr(getattr(step, "p_flat", None), "device", None) or torch.device("cpu")
    counts = torch.tensor([r[1] if r is not None else 0.0 for r in rows], dtype=torch.float64)
    zero = torch.zeros((), device=dev, dtype=torch.float64)
    losses = torch.stack([r[0] if r is not None else zero for r in rows])
    if not sharded:
        vals = losses.cpu().tolist()                                    # single process: the batch losses themselves
    else:
        tab = torch.stack([losses * counts.to(dev), counts.to(dev)], dim=1)      # one upload, one all-reduce, one download
        step.dist.all_reduce(tab, group=step.pg)
        tab = tab.cpu()
        if bool((tab[:, 1] == 0).any()):
            raise RuntimeError("evaluate_loader: a global batch was scored by no rank (the ranks' loaders disagree on the batch order)")
        vals = (tab[:, 0] / tab[:, 1]).tolist()
    return _mean_loss([0.0 if v != v else v for v in vals], len(vals))


def fit(step, train_loader, val_loader, test_loader, weights_path: str, weights_name: str, loss_values_path: Optional[str] = None,
        val_loss_SMA_window: int = 10, validation_loss_count_threshold: int = 5, train_indefinitely: bool = False,
        save_at_epochs: Sequence[int] = (), max_epochs: Optional[int] = None,
        train_pass: Optional[Callable] = None, eval_pass: Optional[Callable] = None, save: Optional[Callable] = None,
        echo: Callable[[str], None] = print) -> Dict[str, List[float]]:
    """Run epochs until the stopping rule fires (or `max_epochs`).  Returns H = {train_loss, validation_loss, test_loss}.

    `train_pass(step, loader) -> (sum_of_batch_losses, n_batches)`, `eval_pass(step, loader) -> mean_loss` and
    `save(step, path)` default to the libgsd implementations; tests substitute host stubs for them.

    Data parallel (a build-side addition, the reference is single-process): every rank runs the same epochs on its shard
    of each batch; the epoch losses are averaged over the ranks (`step.mean_across_ranks`) so that all ranks take the
    same stopping and checkpoint decisions, and only rank 0 writes checkpoints, the log file and the echo.  Validation and
    test passes walk the GLOBAL batches with every rank scoring its own share (`evaluate_loader`): the value is the
    single-process one (no wrap-around padding in the loss early stopping reads), identical on every rank, at the per-rank
    train shape; `across()` of it is then the identity up to the last bit."""
    if train_pass is None:
        from .dataset import train_epoch as train_pass
    if eval_pass is None:
        eval_pass = evaluate_loader
    elif getattr(val_loader, "world_size", 1) > 1 and hasattr(val_loader, "unsharded"):
        # a caller-supplied pass does not know about shares: it gets single-process loaders (no wrap-around-padded shards in the
        # loss early stopping reads); the default pass scores every rank's share of the GLOBAL batches instead
        val_loader, test_loader = val_loader.unsharded(), test_loader.unsharded()
    if save is None:
        def save(st, path):
            st.save_checkpoint(path, use_ema=True)
    H: Dict[str, List[float]] = {"train_loss": [], "validation_loss": [], "test_loss": []}
    stopper = EarlyStopping(val_loss_SMA_window, validation_loss_count_threshold, train_indefinitely)
    is_main = getattr(step, "rank", 0) == 0
    across = getattr(step, "mean_across_ranks", float)
    log = open(loss_values_path, "a") if (loss_values_path and is_main) else None

    def emit(line: str) -> None:
        if not is_main:
            return
        echo(line)
        if log is not None:
            log.write(line + "\n")
    start = time.time()
    try:
        e = 0
        while True:
            t0 = time.time()
            total, nb = train_pass(step, train_loader)
            train_loss = across(total / nb if nb else 0.0)
            H["train_loss"].append(train_loss)
            validation_loss = across(eval_pass(step, val_loader))
            H["validation_loss"].append(validation_loss)
            test_loss = across(eval_pass(step, test_loader))
            H["test_loss"].append(test_loss)
            stop, stalled, new_min = stopper.update(validation_loss)
            if stalled:
                emit(f"Validation loss stopped decreasing at epoch {e + 1}")
            if new_min:
                emit("Validation loss is at a minimum. Saving the model")
                if is_main:
                    os.makedirs(weights_path, exist_ok=True)
                    save(step, os.path.join(weights_path, weights_name + ".pth"))
            if train_indefinitely and len(save_at_epochs) > 0 and e in save_at_epochs and is_main:
                os.makedirs(weights_path, exist_ok=True)
                save(step, os.path.join(weights_path, weights_name + "_epoch" + str(e) + ".pth"))
            emit("[INFO] EPOCH: {}".format(e + 1))
            emit("Train loss: {:.6f},  Validation loss: {:.6f}, Test loss: {:.6f}".format(train_loss, validation_loss, test_loss))
            emit(f"Time for epoch: {time.time() - t0}")
            e += 1
            if stop or (max_epochs is not None and e >= max_epochs):
                break
        emit("Training complete")
        emit("Training time: {}s".format(time.time() - start))
    finally:
        if log is not None:
            log.close()
    return H
