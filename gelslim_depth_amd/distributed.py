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
    """Sum a flat gradient arena across ranks, bucket by bucket."""

    def __init__(self, g_flat: torch.Tensor, buckets: List[Tuple[str, int, int]], group=None, overlap: bool = True,
                 force: bool = False):
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

    def on_block_done(self, tag: str) -> None:
        self._done.append(tag)
        if (self.world > 1 or self.force) and self.overlap and tag in self.buckets:
            lo, hi = self.buckets[tag]
            self._works.append(self.dist.all_reduce(self.g[lo:hi], group=self.group, async_op=True))

    def finish(self) -> None:
        """After backward: every bucket has been summed when this returns (stream-ordered for NCCL)."""
        if self.world > 1 or self.force:
            if self.overlap:
                missing = [t for t in self.buckets if t not in self._done]
                for t in missing:          # a block the schedule did not announce: reduce it now
                    lo, hi = self.buckets[t]
                    self._works.append(self.dist.all_reduce(self.g[lo:hi], group=self.group, async_op=True))
                for w in self._works:
                    w.wait()
            else:
                self.dist.all_reduce(self.g, group=self.group)
        self._works = []
        self._done = []


def broadcast_state(p_flat: torch.Tensor, buffers: Sequence[torch.Tensor], group=None, src: int = 0) -> None:
    """Rank `src`'s parameters and BatchNorm buffers become everyone's (what DDP does at construction)."""
    import torch.distributed as dist
    dist.broadcast(p_flat, src=src, group=group)
    for b in buffers:
        dist.broadcast(b, src=src, group=group)
