#!/usr/bin/env python3
"""bf16 train step (batch 32) with the main chain on a high-priority stream and the side-stream weight gradients cut into more, smaller
blocks (GSD_BF16_WGRAD_BLOCKS): does the dispatcher let the backward chain's kernels in between the side stream's blocks?
usage (GPU box): PYTHONPATH=. python profiles/bench_stream_priority.py"""
import os
import subprocess
import sys
import time

if len(sys.argv) > 1:
    import torch
    from gelslim_depth_amd import synth
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    hp = sys.argv[1] == "1"
    dims = [64, 128, 256, 512, 1024]
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision="bf16").to("cuda").train()
    step = TrainStep(m)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    x = torch.rand((32, 3, 320, 427), device="cuda", generator=g)
    t = torch.rand((32, 1, 320, 427), device="cuda", generator=g)
    print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a", file=sys.stderr)
    s = torch.cuda.Stream(priority=-1) if hp else torch.cuda.current_stream()
    with torch.cuda.stream(s):
        for _ in range(5):
            step(x, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step(x, t)
        torch.cuda.synchronize()
        print(f"{(time.perf_counter() - t0) / 20 * 1e3:.3f}")
    sys.exit(0)

for blocks in ("512", "1024", "2048"):
    for hp in ("0", "1", "0", "1"):
        env = dict(os.environ, GSD_BF16_WGRAD_BLOCKS=blocks)
        r = subprocess.run([sys.executable, __file__, hp], env=env, capture_output=True, text=True)
        print(f"3x3 dW blocks {blocks}, main chain on a high-priority stream: {hp} -> {r.stdout.strip()} ms/step", flush=True)
        if r.returncode != 0:
            print(r.stderr[-2000:])
