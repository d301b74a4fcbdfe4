"""Worker for tests/test_gpu_ddp.py: one rank of a 2-rank data-parallel fused train step.
On a node with at least WORLD_SIZE GPUs every rank takes cuda:LOCAL_RANK and the production backend "nccl" (RCCL over xGMI);
on the one-card GPU box both ranks share cuda:0 over gloo (RCCL refuses two ranks on one device).  Same assertions either
way.  GSD_DDP_BACKEND=gloo forces the shared-card form.  Writes its results to <outdir>/rank<r>.npz."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


class _TensorSet:
    """What DeviceLoader needs from a DeviceDataset, over two resident tensors."""

    def __init__(self, x, t):
        self.x, self.t, self.device = x, t, x.device

    def __len__(self):
        return self.x.shape[0]

    def batch(self, idx):
        return {"tactile_image": self.x[idx], "depth_image": self.t[idx], "object_index": idx}


def main():
    outdir, mode = sys.argv[1], sys.argv[2]          # mode: "local_bn" | "sync_bn" | "eval"
    precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    use_rccl = torch.cuda.device_count() >= world and os.environ.get("GSD_DDP_BACKEND", "nccl") == "nccl"
    dev = torch.device("cuda", local if use_rccl else 0)
    torch.cuda.set_device(dev)
    if use_rccl:
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            del os.environ["NCCL_DEBUG"]
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from gelslim_depth_amd import synth
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    full = len(sys.argv) > 4 and sys.argv[4] == "full"     # BASELINE's network and resolution: the REAL bucket sizes (0.15-57 MB)
    dims = [64, 128, 256, 512, 1024] if full else ([16, 32, 64] if precision == "fp32" else [32, 64, 128])
    if len(sys.argv) > 5:                                   # explicit layer dimensions, e.g. "64,128": the fused `inc` block and the
        dims = [int(v) for v in sys.argv[5].split(",")]     # weights-resident 64 -> 64 kernel of the bf16 engine
    # every rank starts from DIFFERENT weights: the rank-0 broadcast must fix that
    st = synth.make_state(3, 1, dims, 5 + 100 * rank, "conditioned")
    x, t = synth.make_batch(4, 320, 427, 6) if full else synth.make_batch(4, 37, 53, 6)            # global batch 4 -> 2 per rank
    per = 4 // world
    xs, ts = x[rank * per:(rank + 1) * per], t[rank * per:(rank + 1) * per]
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision=precision)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to(dev).train()
    step = TrainStep(m, process_group=dist.group.WORLD, sync_bn=(mode == "sync_bn"), overlap_allreduce=True)
    if mode == "eval":
        # validation pass under data parallelism (harness.evaluate_loader): 7 samples, batch 2 per rank -> global batches of
        # 4 and 3 (rank 1 scores ONE sample of the tail); 5 samples -> 4 and 1 (rank 1 gets nothing of the tail)
        from gelslim_depth_amd import harness
        from gelslim_depth_amd.dataset import DeviceLoader
        out = {"backend": dist.get_backend()}
        for n in (7, 5, 8):
            xe, te = synth.make_batch(n, 37, 53, 9)
            ds = _TensorSet(torch.from_numpy(xe).to(dev), torch.from_numpy(te).to(dev))
            out[f"val{n}"] = harness.evaluate_loader(step, DeviceLoader(ds, batch_size=2, rank=rank, world_size=world))
            out[f"shape{n}"] = np.array(m._engine._shape[:3])
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
        dist.barrier()
        dist.destroy_process_group()
        return
    p0 = step.p_flat.cpu().numpy().copy()
    loss = step(torch.from_numpy(xs).to(dev), torch.from_numpy(ts).to(dev)).item()
    torch.cuda.synchronize()
    out = {"backend": dist.get_backend(), "loss": loss, "p0": p0, "g_sum": step.g_flat.cpu().numpy(), "p1": step.p_flat.cpu().numpy()}
    for k, v in m.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["buf/" + k] = v.cpu().numpy()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
