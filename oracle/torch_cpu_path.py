"""ORACLE / CPU BASELINE (test infrastructure, NOT product code).

The reference is pure Python whose arithmetic is executed by torch's CPU kernels
(/root/reference/gelslim_depth/models/unet.py:11-16,26,36,46-48,54 call sites;
train_utils/train_unet.py:52,306,374-375).  The reference's files cannot travel to the
GPU box, so this file re-types the SAME torch-operator sequence in functional form over a
reference-layout state dict; bench.py times it on the host cores as `cpu_baseline`
(kind "port") and tests use it as a fast second checker next to oracle/unet_numpy.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Pinned by tests/test_oracle.py against tests/golden/*.npz (made by running the reference).
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F


def _n_down(state) -> int:
    return sum(1 for k in state if k.startswith("down.") and k.endswith("double_conv.0.weight"))


def _block(x, s, prefix: str, train: bool):
    # (conv3x3 p1 no-bias -> BN -> ReLU) x2 : unet.py:9-16
    for ci, bi in ((0, 1), (3, 4)):
        x = F.conv2d(x, s[f"{prefix}.double_conv.{ci}.weight"], None, 1, 1)
        p = f"{prefix}.double_conv.{bi}."
        x = F.batch_norm(x, s[p + "running_mean"], s[p + "running_var"], s[p + "weight"], s[p + "bias"],
                         train, 0.1, 1e-5)
        if train:
            s[p + "num_batches_tracked"] += 1
        x = F.relu(x)
    return x


def forward(state: Dict[str, torch.Tensor], x: torch.Tensor, train: bool = False) -> torch.Tensor:
    """unet.py:79-88.  `state` tensors are used in place (BN running stats are updated when train)."""
    nd = _n_down(state)
    feats: List[torch.Tensor] = [_block(x, state, "inc", train)]
    for i in range(nd):
        feats.append(_block(F.max_pool2d(feats[-1], 2), state, f"down.{i}.maxpool_conv.1", train))   # unet.py:26
    cur = feats[-1]
    for i in range(nd):
        skip = feats[-2 - i]
        up = F.conv_transpose2d(cur, state[f"up.{i}.up.weight"], state[f"up.{i}.up.bias"], stride=2)  # unet.py:36,41
        dy, dx = skip.shape[2] - up.shape[2], skip.shape[3] - up.shape[3]
        up = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])                                  # unet.py:46-47
        cur = _block(torch.cat([skip, up], 1), state, f"up.{i}.conv", train)                            # unet.py:48-49
    return F.conv2d(cur, state["outc.conv.weight"], state["outc.conv.bias"])                           # unet.py:54


def is_param(name: str) -> bool:
    return not (name.endswith("running_mean") or name.endswith("running_var")
                or name.endswith("num_batches_tracked"))


class CpuTrainer:
    """Reference step body (train_unet.py:346-347,370,374-375) on device='cpu'."""

    def __init__(self, state_np, lr=1e-3, wd=1e-6):
        self.state = {k: torch.from_numpy(v.copy()) if not isinstance(v, torch.Tensor) else v.clone()
                      for k, v in state_np.items()}
        self.params = [k for k in self.state if is_param(k)]
        for k in self.params:
            self.state[k].requires_grad_(True)
        self.opt = torch.optim.Adam([self.state[k] for k in self.params], lr=lr, weight_decay=wd)

    def step(self, x: torch.Tensor, target: torch.Tensor) -> float:
        self.opt.zero_grad()
        out = forward(self.state, x, train=True)
        loss = torch.mean((out - target) ** 2)      # train_unet.py:51-52
        loss.backward()
        self.opt.step()
        return float(loss.item())

    def grads(self):
        return {k: self.state[k].grad.detach().numpy() for k in self.params}


# ---------------------------------------------------------------------------------------------------------------------
# bf16 mixed-precision EMULATION (BASELINE.json configs[4]).  The reference has no reduced-precision path; this restates
# where gelslim_depth_amd/engine_bf16.py rounds to bfloat16, on top of the same operator sequence, so that the HIP path
# can be checked against something tighter than "within bf16 noise of fp32":
#   * every stored activation (raw conv output y, relu(bn(y)), pooled, upsampled+bias) and every stored activation
#     gradient is rounded to bf16 (straight-through in autograd; gradients are rounded by hooks);
#   * conv / convT weights are rounded to bf16 copies, fp32 accumulation; the 1x1 output conv keeps fp32 weights;
#   * BatchNorm statistics, parameters, parameter gradients, loss: fp32.
# It is an emulation, not a bit-exact model: accumulation order and the exact BatchNorm-backward formulation differ.
# ---------------------------------------------------------------------------------------------------------------------
def _q(t: torch.Tensor) -> torch.Tensor:
    """round to bf16, identity gradient, and round the gradient flowing back through this tensor"""
    r = t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()
    if r.requires_grad:
        r.register_hook(lambda g: g.to(torch.bfloat16).to(g.dtype))
    return r


def _block_bf16(x, s, prefix: str, train: bool):
    for ci, bi in ((0, 1), (3, 4)):
        y = _q(F.conv2d(x, _q(s[f"{prefix}.double_conv.{ci}.weight"]), None, 1, 1))
        p = f"{prefix}.double_conv.{bi}."
        y = F.batch_norm(y, s[p + "running_mean"], s[p + "running_var"], s[p + "weight"], s[p + "bias"], train, 0.1, 1e-5)
        if train:
            s[p + "num_batches_tracked"] += 1
        x = _q(F.relu(y))
    return x


def forward_bf16(state: Dict[str, torch.Tensor], x: torch.Tensor, train: bool = False) -> torch.Tensor:
    nd = _n_down(state)
    feats: List[torch.Tensor] = [_block_bf16(x.to(torch.bfloat16).to(torch.float32), state, "inc", train)]
    for i in range(nd):
        feats.append(_block_bf16(_q(F.max_pool2d(feats[-1], 2)), state, f"down.{i}.maxpool_conv.1", train))
    cur = feats[-1]
    for i in range(nd):
        skip = feats[-2 - i]
        up = _q(F.conv_transpose2d(cur, _q(state[f"up.{i}.up.weight"]), state[f"up.{i}.up.bias"], stride=2))
        dy, dx = skip.shape[2] - up.shape[2], skip.shape[3] - up.shape[3]
        up = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        cur = _block_bf16(torch.cat([skip, up], 1), state, f"up.{i}.conv", train)
    return F.conv2d(cur, state["outc.conv.weight"], state["outc.conv.bias"])


class CpuTrainerBF16(CpuTrainer):
    """CpuTrainer with the bf16 emulation as the forward (fp32 master weights and Adam, as the HIP path)."""

    def step(self, x: torch.Tensor, target: torch.Tensor) -> float:
        self.opt.zero_grad()
        out = forward_bf16(self.state, x, train=True)
        loss = torch.mean((out - target) ** 2)
        loss.backward()
        self.opt.step()
        return float(loss.item())
