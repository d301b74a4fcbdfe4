#!/usr/bin/env python3
"""Feasibility probe: the bf16 train step's forward + loss + backward (178 launches, two streams) replayed as ONE hipGraph, Adam launched
eagerly behind it -- how much of the ~0.8 ms of dispatch gaps does a graph take back?   PYTHONPATH=. python profiles/bench_train_graph.py"""
import time

import torch

from gelslim_depth_amd import _lib as L
from gelslim_depth_amd.models.unet import UNet
from gelslim_depth_amd.train import TrainStep, loss_fwd_bwd

dims = [64, 128, 256, 512, 1024]
m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision="bf16").to("cuda").train()
step = TrainStep(m)
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.rand((32, 3, 320, 427), device="cuda", generator=g)
t = torch.rand((32, 1, 320, 427), device="cuda", generator=g)
for _ in range(5):
    step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step(x, t)
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step")

eng = m._engine
P = m._tensor_map()


def fwd_bwd():
    out = eng.forward(x, P, train=True, out=step._out)
    loss_fwd_bwd(step.loss_kind, out, t, step._dout, step.loss_buf, step.loss_ws, guard=None)
    eng.backward(step._dout, P, m._grad_views)


def adam():
    step.step_count += 1
    L.check(L.lib.gsd_adam_ema(step.p_flat.data_ptr(), step.g_flat.data_ptr(), step.m_flat.data_ptr(), step.v_flat.data_ptr(),
                               L.ptr(step.ema_flat), step.numel, step.step_count, step.lr, step.betas[0], step.betas[1], step.eps,
                               step.wd, 0.0, 1.0, None, L.stream_ptr()), "adam")


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        fwd_bwd()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    fwd_bwd()
torch.cuda.synchronize()
for _ in range(3):
    graph.replay()
    adam()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    graph.replay()
    adam()
torch.cuda.synchronize()
print(f"graph replay of forward + loss + backward, eager Adam: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step   loss {float(step.loss_buf[0]):.5f}")
