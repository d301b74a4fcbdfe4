"""What the data-parallel plumbing costs ONE rank at the per-GPU batch of 8 (fp32): the plain step, the step with the nine
gradient buckets handed to RCCL (one-rank group: everything but the wire), and the step with ONE all-reduce behind the backward.
usage (GPU box): PYTHONPATH=. python profiles/bench_handoff.py [batch]"""
import os
import sys
import time
import torch
import torch.distributed as dist
from gelslim_depth_amd import synth
from gelslim_depth_amd.models.unet import UNet
from gelslim_depth_amd.train import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DIMS, H, W = [64, 128, 256, 512, 1024], 320, 427
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29561")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
st = synth.make_state(3, 1, DIMS, 0, "conditioned")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.rand((B, 3, H, W), device=dev, generator=g)
t = -0.9 * torch.rand((B, 1, H, W), device=dev, generator=g)


def run(tag, **kw):
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=DIMS)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
    m = m.to(dev).train()
    step = TrainStep(m, **kw)
    for _ in range(3):
        step(x, t)
    res = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step(x, t)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 10 * 1e3)
    print("%-58s %s ms/step" % (tag, " ".join("%.3f" % r for r in res)), flush=True)
    del step, m
    torch.cuda.empty_cache()


run("plain step (no process group)")
run("nine buckets handed to RCCL as they become final", process_group=dist.group.WORLD, force_sync=True)
run("one all-reduce of the arena behind the backward", process_group=dist.group.WORLD, force_sync=True, overlap_allreduce=False)
os.environ["GSD_SIDE_DW"] = "0"
run("plain step, weight gradients on the main stream")
run("nine buckets, weight gradients on the main stream", process_group=dist.group.WORLD, force_sync=True)
dist.destroy_process_group()
