"""hipGraph replay of the eval-mode forward (launch-bound at small batch: ~120 kernel launches for 0.5-4 ms of GPU work).

The engine's schedule is static for a fixed input shape: every buffer it touches is allocated once (`_ensure`), every
kernel argument is a device pointer or a shape, nothing synchronises with the host.  So one eager warm-up (which also sets
the kernels' LDS attributes) followed by a capture on a side stream gives a graph whose replay re-reads the CURRENT
parameter values (the per-launch weight re-layout is part of the graph), i.e. it stays valid across optimiser steps and
`load_state_dict` as long as the parameter storages are not replaced.

Not in the reference (eager torch); SURVEY.md section 8(d) config 1 / DESIGN.md "next" item.  Training is not captured:
its Adam step count and EMA decay are host scalars that change every step.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .models.unet import UNet


class GraphedInference:
    """`fn = GraphedInference(model, example); y = fn(x)`.  `y` is a static buffer that the next call overwrites."""

    def __init__(self, model: UNet, example: torch.Tensor, warmup: int = 2) -> None:
        if model.training:
            raise L.GsdError("GraphedInference captures the eval-mode forward: call model.eval() first")
        if not example.is_cuda or example.dtype != torch.float32:
            raise L.GsdError("GraphedInference needs a float32 example on the GPU")
        self.model = model
        self.x = example.detach().clone().contiguous()
        n, _, h, w = self.x.shape
        self.y = torch.empty((n, model.n_classes, h, w), device=self.x.device, dtype=torch.float32)
        self._params = [p.data_ptr() for p in model.parameters()] + [b.data_ptr() for b in model.buffers()]
        eng = model._engine
        side = torch.cuda.Stream(device=self.x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):            # allocations, hipFuncSetAttribute, lazy module loads
                eng.forward(self.x, model._tensor_map(), train=False, out=self.y)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            eng.forward(self.x, model._tensor_map(), train=False, out=self.y)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self.x.shape:
            raise L.GsdError(f"captured for input shape {tuple(self.x.shape)}, got {tuple(x.shape)}")
        now = [p.data_ptr() for p in self.model.parameters()] + [b.data_ptr() for b in self.model.buffers()]
        if now != self._params:
            raise L.GsdError("a parameter or buffer storage was replaced since capture (e.g. a TrainStep moved the parameters "
                             "into its arena): build a new GraphedInference")
        self.x.copy_(x)
        self.graph.replay()
        return self.y
