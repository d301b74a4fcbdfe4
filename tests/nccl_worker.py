"""Worker for tests/test_gpu_ddp.py::test_one_rank_rccl_step_is_bit_equal: ONE rank, backend "nccl" (RCCL) -- the
production backend of the data-parallel step -- with the collectives forced on (force_sync): rank-0 broadcast, the
bucketed asynchronous gradient all-reduce on RCCL's stream overlapped with backward on the compute stream, the SyncBN
all-reduce of the fp64 sums, the guard word all-reduce.  With one rank every collective is the identity, so the result
must equal the step without a process group bit for bit.  Writes <outdir>/nccl.npz."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    outdir, precision, port = sys.argv[1], sys.argv[2], sys.argv[3]
    torch.cuda.set_device(0)
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        del os.environ["NCCL_DEBUG"]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    from gelslim_depth_amd import synth
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    dims = [16, 32, 64] if precision == "fp32" else [32, 64, 128]
    st = synth.make_state(3, 1, dims, 5, "conditioned")
    x, t = synth.make_batch(3, 37, 53, 6)
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision=precision)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to("cuda:0").train()
    # bf16: the SyncBN path reduces the BatchNorm partial rows in two launches instead of one (another fp64 summation order),
    # so bit-equality with the plain step holds for the fp32 engine only; the bf16 run exercises the gradient collectives
    step = TrainStep(m, process_group=dist.group.WORLD, sync_bn=(precision == "fp32"), overlap_allreduce=True,
                     force_sync=True, nan_policy="skip")
    assert step.sync is not None and step.sync.force, "the collectives must run"
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    losses = [float(step(xd, td).item()) for _ in range(3)]
    torch.cuda.synchronize()
    out = {"losses": np.array(losses), "g": step.g_flat.cpu().numpy(), "p": step.p_flat.cpu().numpy(),
           "ema": step.ema_flat.cpu().numpy(), "world": dist.get_world_size(), "backend": dist.get_backend(),
           "skipped": step.skipped_steps()}
    for k, v in m.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["buf/" + k] = v.cpu().numpy()
    np.savez(os.path.join(outdir, "nccl.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
