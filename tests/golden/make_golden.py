#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container, where /root/reference exists (it does not
travel to the GPU box; the .npz files committed next to this script do).
It imports the reference's own model file
    /root/reference/gelslim_depth/models/unet.py  (DoubleConv:7, Down:22, Up:33,
    OutConv:51, UNet:60)
and drives it exactly like the reference trainer does
    /root/reference/train_utils/train_unet.py:51-52 (MSE), :306 (Adam lr 1e-3,
    wd 1e-6), :346-347 (zero_grad, unet(x=...)), :374-375 (backward, step).
Inputs/weights come from the build-owned seeded generators in
gelslim_depth_amd/synth.py, so tests regenerate them instead of shipping them.

A fixture is data only: inputs (or their seeds) and expected outputs.

EMA: the reference uses torch_ema==0.3 (requirements.txt:6, train_unet.py:309,376),
which is not installed here and not vendored; the `ema_*` arrays are produced by
the *published* torch_ema 0.3 update rule restated below and are therefore
"parity unpinned" (flagged as such in the fixture and in DESIGN.md).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from gelslim_depth.models.unet import UNet, DoubleConv, Down, Up, OutConv  # noqa: E402  (reference)
from gelslim_depth_amd import synth  # noqa: E402  (build-owned generators)

torch.set_num_threads(8)
torch.manual_seed(0)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rnd(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape, dtype=np.float32) * np.float32(scale)).astype(np.float32)


def load_into(module, state):
    module.load_state_dict(OrderedDict((k, t(v)) for k, v in state.items()), strict=True)


def sub_state(rng, module):
    """Random, well-conditioned state for a sub-module (per-op fixtures)."""
    st = OrderedDict()
    for k, v in module.state_dict().items():
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            st[k] = np.zeros((), np.int64)
        elif k.endswith("running_var"):
            st[k] = rng.uniform(0.5, 1.5, shp).astype(np.float32)
        elif k.endswith("running_mean"):
            st[k] = rnd(rng, *shp, scale=0.1)
        elif len(shp) == 4:
            fan = shp[1] * shp[2] * shp[3]
            st[k] = rnd(rng, *shp, scale=float(np.sqrt(2.0 / fan)))
        elif k.endswith("weight"):      # BN gamma
            st[k] = rng.uniform(0.5, 1.5, shp).astype(np.float32)
        else:
            st[k] = rng.uniform(-0.2, 0.2, shp).astype(np.float32)
    return st


def grads_of(module):
    return OrderedDict((k, p.grad.detach().numpy().copy()) for k, p in module.named_parameters())


def buffers_of(module):
    return OrderedDict((k, b.detach().numpy().copy()) for k, b in module.named_buffers())


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB, {len(arrays)} arrays")


def pack(prefix, d):
    return {f"{prefix}/{k}": v for k, v in d.items()}


# ----------------------------------------------------------------------------- per-op
def g_op_module(name, module, inputs, seed):
    """Train-mode fwd+bwd with a random upstream gradient, then eval-mode fwd."""
    rng = np.random.Generator(np.random.PCG64(seed))
    st = sub_state(rng, module)
    load_into(module, st)
    xs = [t(a).requires_grad_(True) for a in inputs]
    module.train()
    y = module(*xs)
    dy = rnd(rng, *y.shape)
    y.backward(t(dy))
    out = {}
    out.update(pack("state", st))
    for i, a in enumerate(inputs):
        out[f"in{i}"] = a
        out[f"din{i}"] = xs[i].grad.numpy().copy()
    out["y_train"] = y.detach().numpy().copy()
    out["dy"] = dy
    out.update(pack("grad", grads_of(module)))
    out.update(pack("buf_after", buffers_of(module)))
    module.eval()
    with torch.no_grad():
        out["y_eval"] = module(*[t(a) for a in inputs]).numpy().copy()   # uses UPDATED running stats
    save(name, **out)


def g_ops():
    rng = np.random.Generator(np.random.PCG64(1234))
    g_op_module("gop_doubleconv.npz", DoubleConv(5, 7), [rnd(rng, 2, 5, 9, 11)], 11)
    g_op_module("gop_down.npz", Down(4, 6), [rnd(rng, 2, 4, 9, 11)], 12)
    # Up: x1 2x8x4x5 -> convT -> 2x4x8x10, skip 2x4x9x11 => diffY=diffX=1 (pad right/bottom, unet.py:43-47)
    g_op_module("gop_up.npz", Up(8, 4), [rnd(rng, 2, 8, 4, 5), rnd(rng, 2, 4, 9, 11)], 13)
    # Up with no padding (even sizes) and diff=2/3 to exercise left/top pad
    g_op_module("gop_up_pad23.npz", Up(8, 4), [rnd(rng, 1, 8, 3, 4), rnd(rng, 1, 4, 8, 11)], 14)

    # OutConv + MSE loss (train_unet.py:51-52)
    rng = np.random.Generator(np.random.PCG64(15))
    m = OutConv(6, 1)
    st = sub_state(rng, m)
    load_into(m, st)
    x = rnd(rng, 2, 6, 7, 9)
    tgt = rnd(rng, 2, 1, 7, 9)
    xt = t(x).requires_grad_(True)
    y = m(xt)
    loss = torch.mean((y - t(tgt)) ** 2)
    loss.backward()
    out = {"in0": x, "target": tgt, "y": y.detach().numpy().copy(), "loss": np.float32(loss.item()),
           "din0": xt.grad.numpy().copy()}
    out.update(pack("state", st))
    out.update(pack("grad", grads_of(m)))
    save("gop_outconv_mse.npz", **out)


# ----------------------------------------------------------------------------- whole nets
def ema_update_restated(shadow, params, num_updates, decay=0.995):
    """torch_ema 0.3 ExponentialMovingAverage.update(), restated from its published source
    (use_num_updates=True): d = min(decay, (1+n)/(10+n)) after n += 1;
    shadow -= (1-d) * (shadow - param).  PARITY UNPINNED (package not available here)."""
    num_updates += 1
    d = min(decay, (1.0 + num_updates) / (10.0 + num_updates))
    one_minus = 1.0 - d
    for s, p in zip(shadow, params):
        tmp = (s - p.detach()) * one_minus
        s.sub_(tmp)
    return num_updates


def g_net(name, dims, n, h, w, seed, init, steps=3, store_grads=True, store_params=True):
    st0 = synth.make_state(3, 1, dims, seed, init)
    x, tgt = synth.make_batch(n, h, w, seed + 1)
    net = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, kernel_size=3, maxpool_size=2, upconv_stride=2)
    load_into(net, st0)
    out = {"meta/dims": np.array(dims), "meta/nhw": np.array([n, h, w]), "meta/seed": np.array(seed),
           "meta/init": np.array(init)}

    # eval-mode output with the initial running stats (inference path, test_depth_estimation.py:65,17)
    net.eval()
    with torch.no_grad():
        out["y_eval0"] = net(x=t(x)).numpy().copy()

    # 3 training steps exactly as train_unet.py:346-376 (same batch every step)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-6)
    shadow = [p.detach().clone() for p in net.parameters()]
    n_upd = 0
    losses = []
    for s in range(steps):
        opt.zero_grad()
        y = net(x=t(x))
        loss = torch.mean((y - t(tgt)) ** 2)
        loss.backward()
        if s == 0:
            out["y_train0"] = y.detach().numpy().copy()
            if store_grads:
                out.update(pack("grad0", grads_of(net)))
            else:
                for k, p in net.named_parameters():
                    g = p.grad.detach().double()
                    out[f"gradsum0/{k}"] = np.array([g.sum().item(), g.abs().sum().item(),
                                                     g.pow(2).sum().sqrt().item()])
                    flat = p.grad.detach().reshape(-1)
                    idx = np.linspace(0, flat.numel() - 1, num=min(64, flat.numel())).astype(np.int64)
                    out[f"gradsample0/{k}"] = flat[t(idx)].numpy().copy()
        opt.step()
        n_upd = ema_update_restated(shadow, list(net.parameters()), n_upd)
        losses.append(loss.item())
        if s == 0:
            out.update(pack("buf1", buffers_of(net)))
    out["losses"] = np.array(losses, np.float64)
    if store_params:
        out.update(pack(f"param{steps}", OrderedDict((k, p.detach().numpy().copy())
                                                     for k, p in net.named_parameters())))
        out.update(pack(f"ema{steps}_UNPINNED", OrderedDict((k, s.numpy().copy())
                                                            for (k, _), s in zip(net.named_parameters(), shadow))))
    out.update(pack(f"buf{steps}", buffers_of(net)))
    net.eval()
    with torch.no_grad():
        out[f"y_eval{steps}"] = net(x=t(x)).numpy().copy()
    save(name, **out)


def g_full():
    """Config 1 of BASELINE.json: single 3x320x427 image through the full-size reference U-Net."""
    dims = [64, 128, 256, 512, 1024]
    seed = 2024
    st0 = synth.make_state(3, 1, dims, seed, "conditioned")
    x, tgt = synth.make_batch(1, 320, 427, seed + 1)
    net = UNet(n_channels=3, n_classes=1, layer_dimensions=dims)
    load_into(net, st0)
    out = {"meta/dims": np.array(dims), "meta/nhw": np.array([1, 320, 427]), "meta/seed": np.array(seed)}

    stats = OrderedDict()

    def hook(nm):
        def f(_m, _i, o):
            d = o.detach().double()
            stats[nm] = np.array([d.mean().item(), d.abs().mean().item(), d.pow(2).sum().sqrt().item()])
        return f
    hs = [net.inc.register_forward_hook(hook("inc"))]
    for i, d in enumerate(net.down):
        hs.append(d.register_forward_hook(hook(f"down{i}")))
    for i, u in enumerate(net.up):
        hs.append(u.register_forward_hook(hook(f"up{i}")))
    net.eval()
    with torch.no_grad():
        y = net(x=t(x))
    out["y_eval"] = y.numpy().copy()
    out.update(pack("act_eval", stats))
    stats.clear()

    # one full train step at batch 1 (fwd + MSE + bwd), checksums of every gradient
    net.train()
    y = net(x=t(x))
    out.update(pack("act_train", stats))
    for h_ in hs:
        h_.remove()
    loss = torch.mean((y - t(tgt)) ** 2)
    loss.backward()
    out["y_train"] = y.detach().numpy().astype(np.float16)      # 273 KB; full-precision check uses checksums
    out["y_train_sum"] = np.array([y.detach().double().sum().item(), y.detach().double().abs().sum().item()])
    out["loss"] = np.array(loss.item(), np.float64)
    for k, p in net.named_parameters():
        g = p.grad.detach().double()
        out[f"gradsum/{k}"] = np.array([g.sum().item(), g.abs().sum().item(), g.pow(2).sum().sqrt().item()])
        flat = p.grad.detach().reshape(-1)
        idx = np.linspace(0, flat.numel() - 1, num=min(64, flat.numel())).astype(np.int64)
        out[f"gradsample/{k}"] = flat[t(idx)].numpy().copy()
    for k, b in net.named_buffers():
        if k.endswith("num_batches_tracked"):
            continue
        d = b.detach().double()
        out[f"bufsum/{k}"] = np.array([d.sum().item(), d.abs().sum().item()])
    save("gfull_b1.npz", **out)


def g_proc():
    """Inference pre/post-processing (test_depth_estimation.py:14-20): the reference's normalisers are imported
    (normalization_utils.py); image_utils.py needs torchvision (absent), so its two one-liners are restated here:
    difference image (image_utils.py:6-10) and F.interpolate(..., mode='area') (image_utils.py:12-15)."""
    import torch.nn.functional as F
    from gelslim_depth.processing_utils import normalization_utils as nu       # reference
    rng = np.random.Generator(np.random.PCG64(77))
    img = rng.uniform(0, 255, (2, 3, 41, 55)).astype(np.float32)
    base = rng.uniform(0, 255, (2, 3, 41, 55)).astype(np.float32)
    diff = (t(img) - t(base) + 255.0) / 2.0
    small = F.interpolate(diff, size=(20, 27), mode="area")
    out = {"img": img, "base": base, "diff": diff.numpy(), "small": small.numpy()}
    out["norm_0_255_to_0_1"] = nu.normalize_tactile_image(small, "0_255_to_0_1", 0.9, None).numpy()
    out["norm_0_255_to_-1_1"] = nu.normalize_tactile_image(small, "0_255_to_-1_1", 0.9, None).numpy()
    params = ([10.0, 20.0, 5.0], [240.0, 200.0, 250.0], [120.0, 110.0, 130.0], [40.0, 50.0, 60.0])
    out["norm_mean_std"] = nu.normalize_tactile_image(small, "mean_std", 0.9, params).numpy()
    # ("min_max_to_-1_1" raises TypeError inside the reference itself: list * float at normalization_utils.py:9)
    depth = -0.9 * rng.random((2, 1, 20, 27), dtype=np.float32)
    out["depth_norm"] = depth
    den = nu.denormalize_depth_image(t(depth), "min_max_to_0_-1", 0.9, (-1.9180814027786255, 0.0))
    out["depth_denorm"] = den.numpy()
    out["depth_full"] = F.interpolate(den, size=(41, 55), mode="area").numpy()
    den2 = nu.denormalize_depth_image(t(depth), "mean_std", 0.9, (-2.0, 0.0, -0.7, 0.3))
    out["depth_denorm_mean_std"] = den2.numpy()
    save("gproc.npz", **out)


def g_dataset():
    """Dataset path (general_dataset.py): GeneralDataset imports torchvision (absent) so the class glue comes from
    oracle/dataset_ref.py (a restatement of its text) -- but run HERE with the reference's own normalisers
    (normalization_utils.py, importable) injected, so the normalised samples are the reference's arithmetic."""
    import torch
    from gelslim_depth.processing_utils import normalization_utils as nu       # reference
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import dataset_ref as dr
    out = {}
    for tag, kw in (("a", dict(use_difference_image=True, image_normalization_method="0_255_to_0_1",
                               depth_normalization_method="min_max_to_0_-1", norm_scale=0.9,
                               max_datapoints_per_object=5)),
                    ("b", dict(use_difference_image=False, image_normalization_method="mean_std",
                               depth_normalization_method="mean_std", norm_scale=1.0, separate_fingers=False))):
        torch.manual_seed(7)
        ds = dr.DatasetOracle(dr.synthetic_objects(11, [3, 4]), dr.synthetic_objects(12, [2]),
                              normalizers=(nu.normalize_tactile_image, nu.normalize_depth_image), **kw)
        out[tag + "_tactile_raw"] = ds.entire_dataset["tactile_image"].numpy()
        out[tag + "_depth_raw"] = ds.entire_dataset["depth_image"].numpy()
        out[tag + "_object_index"] = ds.entire_dataset["object_index"].numpy()
        out[tag + "_depth_params"] = np.array(ds.depth_normalization_parameters, np.float64)
        out[tag + "_image_params"] = np.array(ds.image_normalization_parameters, np.float64)
        samples = [ds[i] for i in range(len(ds))]
        out[tag + "_tactile"] = np.stack([s["tactile_image"].numpy() for s in samples])
        out[tag + "_depth"] = np.stack([s["depth_image"].numpy() for s in samples])
        torch.manual_seed(21)
        out[tag + "_order"] = np.concatenate([b.numpy() for b in dr.loader_order(len(ds), 4)])
    save("gdataset.npz", **out)


if __name__ == "__main__":
    if "--only-proc" in sys.argv:
        g_proc()
        sys.exit(0)
    if "--only-dataset" in sys.argv:
        g_dataset()
        sys.exit(0)
    g_proc()
    g_dataset()
    g_ops()
    # G-tiny: both inits; 21x27 exercises H and W padding in Up (diff=1 both)
    g_net("gtiny_conditioned.npz", [4, 8, 16], 2, 21, 27, 101, "conditioned")
    g_net("gtiny_refinit.npz", [4, 8, 16], 2, 21, 27, 102, "reference")
    # G-mid: channel counts that fill whole MFMA tiles, odd spatial sizes, 4 levels
    g_net("gmid_conditioned.npz", [16, 32, 64, 128], 2, 45, 61, 103, "conditioned",
          store_grads=False, store_params=False)
    g_full()
