"""CPU, world_size 2, gloo: the data-parallel plumbing of the fused train step (gelslim_depth_amd/distributed.py)
-- bucket construction over the real parameter layout, bucketed async all-reduce in backward order, rank-0
broadcast.  No kernels are launched here; the GPU counterpart is tests/test_gpu_ddp.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gelslim_depth_amd import synth
from gelslim_depth_amd.distributed import GradSync, broadcast_state, make_buckets

DIMS = [64, 128, 256, 512, 1024]


def _layout(dims):
    shapes = synth.unet_state_shapes(3, 1, dims)
    names = synth.param_names(list(shapes.keys()))
    offsets, off = {}, 0
    for n in names:
        sz = int(np.prod(shapes[n])) if shapes[n] else 1
        offsets[n] = (off, sz)
        off += sz
    return names, offsets, off


def test_buckets_partition_the_arena_in_backward_order():
    names, offsets, total = _layout(DIMS)
    assert total == 31037633
    b = make_buckets(names, offsets, len(DIMS) - 1)
    assert [t for t, _, _ in b] == ["dec3", "dec2", "dec1", "dec0", "enc4", "enc3", "enc2", "enc1", "enc0"]
    # decoder buckets sit at the end of the arena (last first), encoder buckets before them, no gaps/overlaps
    spans = sorted((lo, hi) for _, lo, hi in b)
    assert spans[0][0] == 0 and spans[-1][1] == total
    for (l0, h0), (l1, h1) in zip(spans, spans[1:]):
        assert h0 == l1
    d = {t: (lo, hi) for t, lo, hi in b}
    assert d["dec3"][1] == total                      # outc + up.3 finish first and are the arena's tail
    assert d["enc0"][0] == 0
    # tiny net, single level
    n2, o2, t2 = _layout([8])
    b2 = make_buckets(n2, o2, 0)
    assert b2 == [("enc0", 0, t2)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, overlap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dims = [4, 8, 16]
        names, offsets, total = _layout(dims)
        buckets = make_buckets(names, offsets, len(dims) - 1)
        rng = np.random.default_rng(100 + rank)
        g_local = torch.from_numpy(rng.standard_normal(total).astype(np.float32))
        g = g_local.clone()
        sync = GradSync(g, buckets, group=None, overlap=overlap)
        for tag, _, _ in buckets:          # the order UNetEngine.backward announces blocks
            sync.on_block_done(tag)
        sync.finish()
        # expected: elementwise sum over ranks
        exp = sum(torch.from_numpy(np.random.default_rng(100 + r).standard_normal(total).astype(np.float32))
                  for r in range(world))
        ok_sum = torch.allclose(g, exp, rtol=0, atol=1e-6)
        # a second step reuses the object
        g.copy_(g_local)
        sync.on_block_done(buckets[0][0])  # only one block announced: finish() must still reduce everything
        sync.finish()
        ok_missing = torch.allclose(g, exp, rtol=0, atol=1e-6)
        # broadcast: rank 0's parameters/buffers win
        p = torch.full((total,), float(rank + 1))
        bufs = [torch.full((5,), float(rank + 1)), torch.tensor(rank + 7, dtype=torch.long)]
        broadcast_state(p, bufs, group=None)
        ok_bc = bool((p == 1).all()) and bool((bufs[0] == 1).all()) and int(bufs[1]) == 7
        q.put((rank, ok_sum, ok_missing, ok_bc))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_gradsync_world2_gloo(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_sum, ok_missing, ok_bc in res:
        assert ok_sum and ok_missing and ok_bc, (rank, ok_sum, ok_missing, ok_bc)


class _StubDataset:
    """What DeviceLoader needs from a DeviceDataset: len, device, batch(idx)."""
    device = torch.device("cpu")

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def batch(self, idx):
        return {"tactile_image": idx.clone(), "depth_image": idx.clone(), "object_index": idx.clone()}


@pytest.mark.parametrize("n,bs,world", [(65, 32, 2), (7, 2, 2), (9, 2, 4), (3, 4, 2), (64, 32, 2)])
def test_device_loader_gives_every_rank_the_same_number_of_equal_batches(n, bs, world):
    """Every TrainStep issues collectives, so ranks must agree on the number of steps and on the shard size of each
    (n=65, batch 32 x 2 ranks used to leave rank 1 one step short)."""
    from gelslim_depth_amd.dataset import DeviceLoader
    per_rank = []
    for r in range(world):
        torch.manual_seed(11)
        ld = DeviceLoader(_StubDataset(n), batch_size=bs, shuffle=True, rank=r, world_size=world)
        per_rank.append([b["tactile_image"] for b in ld])
        assert len(per_rank[-1]) == len(ld)
    counts = {len(b) for b in per_rank}
    assert len(counts) == 1
    for i in range(len(per_rank[0])):
        sizes = {int(b[i].numel()) for b in per_rank}
        assert len(sizes) == 1 and sizes.pop() > 0
    # together the ranks cover every sample; the padding only repeats samples from the start of the permutation
    torch.manual_seed(11)
    order = DeviceLoader(_StubDataset(n), batch_size=bs, shuffle=True).order()
    seen = torch.cat([per_rank[r][i] for i in range(len(per_rank[0])) for r in range(world)])
    assert torch.equal(seen[:n], order)
    extra = seen[n:]
    assert extra.numel() < world and torch.equal(extra, order[torch.arange(extra.numel()) % n])
    # drop_last: no ragged batch at all
    for r in range(world):
        ld = DeviceLoader(_StubDataset(n), batch_size=bs, shuffle=False, drop_last=True, rank=r, world_size=world)
        assert [int(b["tactile_image"].numel()) for b in ld] == [bs] * (n // (bs * world))


@pytest.mark.parametrize("n,bs,world", [(65, 32, 2), (7, 2, 2), (9, 2, 4)])
def test_unsharded_loader_is_the_single_process_loader(n, bs, world):
    """Validation / test passes (harness.fit) read the sharded loader's UNSHARDED form: the global batches one process would
    see, every sample exactly once -- no wrap-around padding in the loss that early stopping reads."""
    from gelslim_depth_amd.dataset import DeviceLoader
    ref = [b["tactile_image"] for b in DeviceLoader(_StubDataset(n), batch_size=bs * world, shuffle=False)]
    for r in range(world):
        ld = DeviceLoader(_StubDataset(n), batch_size=bs, shuffle=False, rank=r, world_size=world).unsharded()
        got = [b["tactile_image"] for b in ld]
        assert len(got) == len(ref) and all(torch.equal(a, b) for a, b in zip(got, ref))
        assert torch.equal(torch.cat(got), torch.arange(n))
    one = DeviceLoader(_StubDataset(n), batch_size=bs)
    assert one.unsharded() is one


@pytest.mark.parametrize("n,bs,world", [(65, 32, 2), (7, 2, 2), (9, 2, 4), (5, 2, 2), (3, 4, 2), (64, 32, 2)])
def test_eval_shares_cover_every_sample_once_at_the_train_shape(n, bs, world):
    """Validation / test passes under data parallelism (harness.evaluate_loader): rank r scores samples [r*bs, (r+1)*bs) of each
    GLOBAL batch -- no wrap-around, nothing scored twice, every forward at the per-rank train batch size (a short share is
    padded by repeating its last sample and only `valid` samples count; an empty share is skipped)."""
    from gelslim_depth_amd.dataset import DeviceLoader
    ref = [b["tactile_image"] for b in DeviceLoader(_StubDataset(n), batch_size=bs * world, shuffle=False)]
    walks = [list(DeviceLoader(_StubDataset(n), batch_size=bs, shuffle=False, rank=r, world_size=world).eval_shares())
             for r in range(world)]
    assert all(len(w) == len(ref) for w in walks)
    for b, g in enumerate(ref):
        scored = []
        for r in range(world):
            data, valid, count = walks[r][b]
            assert count == g.numel()
            if data is None:
                assert valid == 0
                continue
            assert data["tactile_image"].numel() == bs and 0 < valid <= bs
            scored.append(data["tactile_image"][:valid])
        assert torch.equal(torch.cat(scored), g)
