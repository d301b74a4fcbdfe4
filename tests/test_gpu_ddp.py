"""GPU: the 2-rank data-parallel fused train step (bucketed gradient all-reduce overlapped with backward,
rank-0 broadcast, optional SyncBN) against the oracle.

  * local BN (what torch DDP does): summed gradient arena / 2 == mean of the oracle's per-shard gradients;
  * SyncBN: equals the oracle's SINGLE-process step on the whole global batch;
both ranks end with identical parameters."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, rel_l1
from gelslim_depth_amd import synth

pytestmark = pytest.mark.gpu


def run_two_ranks(tmp_path, mode, precision="fp32", size="small", dims=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(REPO, "tests", "ddp_worker.py"), str(tmp_path), mode, precision, size]
    if dims is not None:
        cmd.append(",".join(str(d) for d in dims))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [dict(np.load(os.path.join(tmp_path, f"rank{i}.npz"))) for i in range(2)]
    import torch
    want = "nccl" if (torch.cuda.device_count() >= 2 and os.environ.get("GSD_DDP_BACKEND", "nccl") == "nccl") else "gloo"
    assert all(str(r["backend"]) == want for r in res), "two GPUs or more: the ranks must have met over RCCL"
    return res


def oracle_flat(grads, names):
    return np.concatenate([grads[n].reshape(-1) for n in names])


@pytest.mark.parametrize("mode", ["local_bn", "sync_bn"])
def test_two_rank_step_matches_oracle(tmp_path, mode):
    from oracle import unet_numpy as on
    res = run_two_ranks(tmp_path, mode)
    dims = [16, 32, 64]
    st0 = synth.make_state(3, 1, dims, 5, "conditioned")          # rank 0's weights are the truth
    names = synth.param_names(list(st0.keys()))
    p0 = oracle_flat(st0, names)
    for r in res:
        assert np.array_equal(r["p0"], p0), "rank-0 broadcast"
    assert np.array_equal(res[0]["g_sum"], res[1]["g_sum"]), "all ranks hold the same summed gradients"
    assert np.array_equal(res[0]["p1"], res[1]["p1"]), "replicas stay in lock-step"
    x, t = synth.make_batch(4, 37, 53, 6)
    if mode == "local_bn":
        gs, ls = [], []
        for r in range(2):
            net, losses, first, _ = on.train_steps(st0, x[2 * r:2 * r + 2], t[2 * r:2 * r + 2], 1)
            gs.append(oracle_flat(first["grads"], names))
            ls.append(losses[0])
        g_exp = (gs[0] + gs[1]) / 2
        for r in range(2):
            assert abs(res[r]["loss"] - ls[r]) < 1e-4 * ls[r]
        p_exp = None
    else:
        net, losses, first, _ = on.train_steps(st0, x, t, 1)
        g_exp = oracle_flat(first["grads"], names)
        p_exp = oracle_flat(net.s, names)
        for k, v in first["buf"].items():
            if not k.endswith("num_batches_tracked"):
                assert rel_l1(res[0]["buf/" + k], v) < 1e-4, k      # running stats from GLOBAL batch statistics
    assert rel_l1(res[0]["g_sum"] / 2, g_exp) < 2e-2
    if p_exp is not None:
        assert rel_l1(res[0]["p1"], p_exp) < 1e-3


def test_two_rank_bf16_step_is_the_sum_of_its_shards(tmp_path):
    """bf16 engine under data parallelism (local BatchNorm statistics): every rank's kernels are deterministic, so the
    all-reduced gradient arena must equal, bit for bit, the fp32 sum of two single-process bf16 steps on the two shards;
    the replicas stay in lock-step and start from rank 0's weights."""
    import torch
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    res = run_two_ranks(tmp_path, "local_bn", "bf16")
    dims = [32, 64, 128]
    st0 = synth.make_state(3, 1, dims, 5, "conditioned")
    x, t = synth.make_batch(4, 37, 53, 6)
    assert np.array_equal(res[0]["g_sum"], res[1]["g_sum"]) and np.array_equal(res[0]["p1"], res[1]["p1"])
    shards = []
    for r in range(2):
        m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision="bf16")
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st0.items()}, strict=True)
        m = m.to("cuda").train()
        step = TrainStep(m)
        loss = float(step(torch.from_numpy(x[2 * r:2 * r + 2]).cuda(), torch.from_numpy(t[2 * r:2 * r + 2]).cuda()))
        assert abs(loss - float(res[r]["loss"])) <= 1e-6 * abs(loss)
        shards.append(step.g_flat.cpu().numpy())
    assert np.array_equal(res[0]["g_sum"], shards[0] + shards[1])


def test_two_rank_bf16_syncbn_matches_single_process_global_batch(tmp_path):
    """bf16 + SyncBN: two ranks with all-reduced BatchNorm sums == one process on the whole batch, up to the rounding
    noise of a different summation split (statistics agree to fp64 round-off, so only isolated bf16 roundings flip)."""
    import torch
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    res = run_two_ranks(tmp_path, "sync_bn", "bf16")
    dims = [32, 64, 128]
    st0 = synth.make_state(3, 1, dims, 5, "conditioned")
    x, t = synth.make_batch(4, 37, 53, 6)
    assert np.array_equal(res[0]["p1"], res[1]["p1"])
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision="bf16")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st0.items()}, strict=True)
    m = m.to("cuda").train()
    step = TrainStep(m)
    loss = float(step(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()))
    g1 = step.g_flat.cpu().numpy()
    g2 = res[0]["g_sum"] / 2.0          # each rank's loss is the mean over ITS half: the sum is twice the global-mean gradient
    assert abs(0.5 * (float(res[0]["loss"]) + float(res[1]["loss"])) - loss) <= 2e-3 * abs(loss)
    cos = float(g1.astype(np.float64) @ g2.astype(np.float64) / np.sqrt((g1.astype(np.float64) ** 2).sum() * (g2.astype(np.float64) ** 2).sum()))
    assert cos > 0.995, cos
    for k, v in m.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_l1(res[0]["buf/" + k], v.cpu().numpy()) < 1e-3, k


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_one_rank_rccl_step_is_bit_equal(tmp_path, precision):
    """The RCCL path on hardware (the GPU box has one card, so one rank): backend "nccl", collectives forced on.  Three
    steps with bucketed async all-reduce + SyncBN + guard exchange == three steps without a process group, bit for bit."""
    import socket
    import torch
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(REPO, "tests", "nccl_worker.py"), str(tmp_path), precision, str(port)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = dict(np.load(os.path.join(tmp_path, "nccl.npz")))
    assert int(res["world"]) == 1 and str(res["backend"]) == "nccl" and int(res["skipped"]) == 0
    dims = [16, 32, 64] if precision == "fp32" else [32, 64, 128]
    st = synth.make_state(3, 1, dims, 5, "conditioned")
    x, t = synth.make_batch(3, 37, 53, 6)
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision=precision)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to("cuda").train()
    step = TrainStep(m)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    losses = [float(step(xd, td).item()) for _ in range(3)]
    assert np.array_equal(np.array(losses), res["losses"])
    assert np.array_equal(step.g_flat.cpu().numpy(), res["g"])
    assert np.array_equal(step.p_flat.cpu().numpy(), res["p"])
    assert np.array_equal(step.ema_flat.cpu().numpy(), res["ema"])
    for k, v in m.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert np.array_equal(v.cpu().numpy(), res["buf/" + k]), k


def test_two_rank_full_size_step_is_the_sum_of_its_shards(tmp_path):
    """The data-parallel step on BASELINE's network at 3x320x427 (2 + 2 images, local BatchNorm statistics): the nine gradient
    buckets are the real ones (0.15 - 57 MB, 124 MB in all).  Every rank's kernels are deterministic, so the all-reduced
    arena must equal, bit for bit, the fp32 sum of two single-process steps on the two shards; both ranks start from rank 0's
    weights and end with identical parameters, equal to a single-process Adam step on the averaged gradient."""
    import torch
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    res = run_two_ranks(tmp_path, "local_bn", "fp32", "full")
    dims = [64, 128, 256, 512, 1024]
    st0 = synth.make_state(3, 1, dims, 5, "conditioned")
    x, t = synth.make_batch(4, 320, 427, 6)
    assert np.array_equal(res[0]["g_sum"], res[1]["g_sum"]) and np.array_equal(res[0]["p1"], res[1]["p1"])
    names = synth.param_names(list(st0.keys()))
    assert np.array_equal(res[1]["p0"], oracle_flat(st0, names)), "rank-0 broadcast"
    shards = []
    for r in range(2):
        m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st0.items()}, strict=True)
        m = m.to("cuda").train()
        step = TrainStep(m)
        loss = float(step(torch.from_numpy(x[2 * r:2 * r + 2]).cuda(), torch.from_numpy(t[2 * r:2 * r + 2]).cuda()))
        assert loss == float(res[r]["loss"])
        shards.append(step.g_flat.cpu().numpy())
        del m, step
        torch.cuda.empty_cache()
    assert np.array_equal(res[0]["g_sum"], shards[0] + shards[1])
    assert res[0]["g_sum"].size == 31037633


def test_two_rank_validation_pass_is_the_single_process_value(tmp_path):
    """harness.evaluate_loader under data parallelism: every rank scores its own share of each GLOBAL batch at the per-rank train
    shape (ragged tails: a short share, an empty share) and one all-reduce of (loss sum, count) per batch gives both ranks
    the value ONE process gets from the unsharded loader -- the same float on both ranks."""
    import torch
    from gelslim_depth_amd import harness
    from gelslim_depth_amd.dataset import DeviceLoader
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    from ddp_worker import _TensorSet
    res = run_two_ranks(tmp_path, "eval")
    dims = [16, 32, 64]
    st0 = synth.make_state(3, 1, dims, 5, "conditioned")
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st0.items()}, strict=True)
    m = m.to("cuda").train()
    step = TrainStep(m)
    for n in (7, 5, 8):
        xe, te = synth.make_batch(n, 37, 53, 9)
        ds = _TensorSet(torch.from_numpy(xe).cuda(), torch.from_numpy(te).cuda())
        one = harness.evaluate_loader(step, DeviceLoader(ds, batch_size=4))        # the global batches in one process
        assert float(res[0][f"val{n}"]) == float(res[1][f"val{n}"]), "every rank takes the same early-stopping decision"
        assert abs(float(res[0][f"val{n}"]) - one) <= 2e-6 * abs(one), (n, float(res[0][f"val{n}"]), one)
        for r in range(2):
            assert tuple(res[r][f"shape{n}"]) == (2, 37, 53), "evaluation keeps the per-rank train shape"


@pytest.mark.parametrize("mode", ["local_bn", "sync_bn"])
def test_two_rank_bf16_with_the_fused_inc_block(tmp_path, mode):
    """The bf16 engine's round-4 kernels under data parallelism (layer_dimensions starting at 64: the `inc` double convolution
    without its first raw output, its recomputing backward, the weights-resident 64 -> 64 kernel).  local BatchNorm: the
    all-reduced arena equals, bit for bit, the sum of two single-process steps on the shards.  SyncBN: the statistics-only pass and
    the fused kernel's partial sums go through the all-reduce like any other layer's; the result tracks one process on the whole
    batch."""
    import torch
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    dims = [64, 128]
    res = run_two_ranks(tmp_path, mode, "bf16", "small", dims=dims)
    st0 = synth.make_state(3, 1, dims, 5, "conditioned")
    x, t = synth.make_batch(4, 37, 53, 6)
    assert np.array_equal(res[0]["g_sum"], res[1]["g_sum"]) and np.array_equal(res[0]["p1"], res[1]["p1"])

    def single(xs, ts):
        m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision="bf16")
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st0.items()}, strict=True)
        m = m.to("cuda").train()
        step = TrainStep(m)
        loss = float(step(torch.from_numpy(xs).cuda(), torch.from_numpy(ts).cuda()))
        assert m._engine.fused_inc and m._engine.c64
        return loss, step.g_flat.cpu().numpy(), m
    if mode == "local_bn":
        shards = [single(x[2 * r:2 * r + 2], t[2 * r:2 * r + 2]) for r in range(2)]
        for r in range(2):
            assert abs(shards[r][0] - float(res[r]["loss"])) <= 1e-6 * abs(shards[r][0])
        assert np.array_equal(res[0]["g_sum"], shards[0][1] + shards[1][1])
    else:
        loss, g1, m = single(x, t)
        g2 = res[0]["g_sum"] / 2.0
        assert abs(0.5 * (float(res[0]["loss"]) + float(res[1]["loss"])) - loss) <= 2e-3 * abs(loss)
        cos = float(g1.astype(np.float64) @ g2.astype(np.float64) / np.sqrt((g1.astype(np.float64) ** 2).sum() * (g2.astype(np.float64) ** 2).sum()))
        assert cos > 0.99, cos
        for k, v in m.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                assert rel_l1(res[0]["buf/" + k], v.cpu().numpy()) < 1e-3, k
