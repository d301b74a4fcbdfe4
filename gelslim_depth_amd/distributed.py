"""Data-parallel plumbing for the fused train step: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for CPU tests).

The reference has no distributed code at all; this is the build-side addition
BASELINE.json asks for: the batch is sharded across ranks, every rank holds the full 124 MB model, and
the only exchange per step is the sum of the flat gradient arena (plus, optionally, the per-channel
BatchNorm sums for SyncBN).  Backward finishes parameter blocks from the END of the arena towards its
start (outc, up.3 ... up.0, down.3 ... inc), so each block is one contiguous range that can be
all-reduced as soon as the block is done, overlapping the rest of backward.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch


def make_buckets(names: Sequence[str], offsets: Dict[str, Tuple[int, int]], n_levels: int) -> List[Tuple[str, int, int]]:
    """[(tag, lo, hi)] in the order backward completes them; tags match UNetEngine.block_done_cb."""
    def rng(prefixes):
        sel = [offsets[n] for n in names if any(n.startswith(p) for p in prefixes)]
        return min(o for o, _ in sel), max(o + s for o, s in sel)

    blocks: List[Tuple[str, int, int]] = []
    for j in reversed(range(n_levels)):
        pref = [f"up.{j}."] + (["outc."] if j == n_levels - 1 else [])
        blocks.append((f"dec{j}",) + rng(pref))
    for lvl in reversed(range(n_levels + 1)):
        pref = ["inc."] if lvl == 0 else [f"down.{lvl - 1}."]
        if n_levels == 0:
            pref.append("outc.")
        blocks.append((f"enc{lvl}",) + rng(pref))
    return blocks


class GradSync:
    """Sum a flat gradient arena across ranks, bucket by bucket.

    timing=True (bench.py --gpus N): every bucket's all-reduce is issued from a side stream of this class between two HIP
    events -- the side stream first waits for the compute stream (the bucket's gradients are final), and `work.wait()` makes
    it wait for RCCL's own stream, so the events bracket exactly the collective (plus its queueing behind earlier buckets).
    `pop_timing()` then says how long the collectives ran per step and how much of that the compute stream had to wait for
    at the end of backward (the part that was NOT overlapped)."""

    def __init__(self, g_flat: torch.Tensor, buckets: List[Tuple[str, int, int]], group=None, overlap: bool = True,
                 force: bool = False, timing: bool = False):
        import torch.distributed as dist
        self.dist = dist
        self.g = g_flat
        self.buckets = {tag: (lo, hi) for tag, lo, hi in buckets}
        self.group = group
        self.overlap = overlap
        self.world = dist.get_world_size(group)
        self.force = force or bool(os.environ.get("GSD_FORCE_SYNC"))   # exercise the collectives with one rank
        self._works: List = []
        self._done: List[str] = []
        self.timing = False
        self._side = None
        self._ev: List[Tuple[str, int, torch.cuda.Event, torch.cuda.Event]] = []     # (tag, bytes, start, end) on the side stream
        self._exposed: List[Tuple[torch.cuda.Event, torch.cuda.Event]] = []          # compute stream: around the final wait
        self.set_timing(timing)

    def set_timing(self, on: bool) -> None:
        """Switch the instrumented path on or off between steps (bench.py times the metric on the production path and the
        collectives in a short pass of their own behind it)."""
        self.timing = bool(on) and self.g.is_cuda
        if self.timing and self._side is None:
            self._side = torch.cuda.Stream(device=self.g.device)
        self._ev, self._exposed = [], []

    def _reduce(self, tag: str, lo: int, hi: int) -> None:
        if not self.timing:
            self._works.append(self.dist.all_reduce(self.g[lo:hi], group=self.group, async_op=True))
            return
        cur = torch.cuda.current_stream(self.g.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            w = self.dist.all_reduce(self.g[lo:hi], group=self.group, async_op=True)
            w.wait()                      # stream-level for RCCL: the side stream waits for the collective's stream
            e1.record()
        self._ev.append((tag, 4 * (hi - lo), e0, e1))

    def on_block_done(self, tag: str) -> None:
        self._done.append(tag)
        if (self.world > 1 or self.force) and self.overlap and tag in self.buckets:
            lo, hi = self.buckets[tag]
            self._reduce(tag, lo, hi)

    def finish(self) -> None:
        """After backward: every bucket has been summed when this returns (stream-ordered for NCCL)."""
        if self.world > 1 or self.force:
            if self.overlap:
                missing = [t for t in self.buckets if t not in self._done]
                for t in missing:          # a block the schedule did not announce: reduce it now
                    lo, hi = self.buckets[t]
                    self._reduce(t, lo, hi)
                for w in self._works:
                    w.wait()
            elif self.timing:
                self._reduce("all", 0, self.g.numel())
            else:
                self.dist.all_reduce(self.g, group=self.group)
            if self.timing:
                cur = torch.cuda.current_stream(self.g.device)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(cur)
                cur.wait_stream(self._side)
                b.record(cur)
                self._exposed.append((a, b))
        self._works = []
        self._done = []

    def pop_timing(self) -> Dict[str, object]:
        """Collective time since the last call (synchronise first): total ms on the side stream, ms the compute stream waited
        for it after backward, bytes, and the per-bucket totals in backward order."""
        per: Dict[str, List[float]] = {}
        total_ms, total_bytes = 0.0, 0
        for tag, nbytes, e0, e1 in self._ev:
            ms = e0.elapsed_time(e1)
            d = per.setdefault(tag, [0.0, 0])
            d[0] += ms
            d[1] += nbytes
            total_ms += ms
            total_bytes += nbytes
        exposed = sum(a.elapsed_time(b) for a, b in self._exposed)
        steps = max(1, len(self._exposed))
        self._ev, self._exposed = [], []
        return {"steps": steps, "allreduce_ms": total_ms, "exposed_ms": exposed, "bytes": total_bytes,
                "buckets": {k: {"ms": round(v[0] / steps, 4), "MB": round(v[1] / steps / 1e6, 3)} for k, v in per.items()}}


def broadcast_state(p_flat: torch.Tensor, buffers: Sequence[torch.Tensor], group=None, src: int = 0) -> None:
    """Rank `src`'s parameters and BatchNorm buffers become everyone's (what DDP does at construction)."""
    import torch.distributed as dist
    dist.broadcast(p_flat, src=src, group=group)
    for b in buffers:
        dist.broadcast(b, src=src, group=group)
