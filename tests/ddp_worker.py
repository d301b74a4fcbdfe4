"""Worker for tests/test_gpu_ddp.py: one rank of a 2-rank data-parallel fused train step.
Both ranks use cuda:0 (the GPU box has one card) and the gloo backend; the production backend is
"nccl" (RCCL), which needs one GPU per rank.  Writes its results to <outdir>/rank<r>.npz."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    outdir, mode = sys.argv[1], sys.argv[2]          # mode: "local_bn" | "sync_bn"
    precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gelslim_depth_amd import synth
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    full = len(sys.argv) > 4 and sys.argv[4] == "full"     # BASELINE's network and resolution: the REAL bucket sizes (0.15-57 MB)
    dims = [64, 128, 256, 512, 1024] if full else ([16, 32, 64] if precision == "fp32" else [32, 64, 128])
    # every rank starts from DIFFERENT weights: the rank-0 broadcast must fix that
    st = synth.make_state(3, 1, dims, 5 + 100 * rank, "conditioned")
    x, t = synth.make_batch(4, 320, 427, 6) if full else synth.make_batch(4, 37, 53, 6)            # global batch 4 -> 2 per rank
    per = 4 // world
    xs, ts = x[rank * per:(rank + 1) * per], t[rank * per:(rank + 1) * per]
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision=precision)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to("cuda:0").train()
    step = TrainStep(m, process_group=dist.group.WORLD, sync_bn=(mode == "sync_bn"), overlap_allreduce=True)
    p0 = step.p_flat.cpu().numpy().copy()
    loss = step(torch.from_numpy(xs).cuda(), torch.from_numpy(ts).cuda()).item()
    torch.cuda.synchronize()
    out = {"loss": loss, "p0": p0, "g_sum": step.g_flat.cpu().numpy(), "p1": step.p_flat.cpu().numpy()}
    for k, v in m.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["buf/" + k] = v.cpu().numpy()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
