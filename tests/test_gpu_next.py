"""GPU parity for the SURVEY 8(f) 'next' rows: N1 inference pre/post-processing, N2 EMA-weights evaluation without
parameter copies, N3 checkpoint written by the fused step."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l1
from gelslim_depth_amd import synth

pytestmark = pytest.mark.gpu


def test_area_resize_affine_vs_golden_and_oracle():
    from gelslim_depth_amd import processing as pp
    from oracle import processing_ref as pr
    g = load_golden("gproc.npz")
    img, base = torch.from_numpy(g["img"]).cuda(), torch.from_numpy(g["base"]).cuda()
    assert rel_l1(pp.get_difference_image(img, base).cpu().numpy(), g["diff"]) < 1e-6
    diff = torch.from_numpy(g["diff"]).cuda()
    assert rel_l1(pp.sample_multi_channel_image_to_desired_size(diff, (20, 27)).cpu().numpy(), g["small"]) < 1e-6
    params = ([10.0, 20.0, 5.0], [240.0, 200.0, 250.0], [120.0, 110.0, 130.0], [40.0, 50.0, 60.0])
    for m, p in (("0_255_to_0_1", None), ("0_255_to_-1_1", None), ("mean_std", params)):
        A, B = pp.tactile_affine(m, 0.9, p)
        got = pp.area_resize_affine(img, (20, 27), A, B, base=base).cpu().numpy()      # difference + resize + normalise fused
        assert rel_l1(got, g["norm_" + m]) < 1e-5, m
    dA, dB = pp.depth_denorm_affine("min_max_to_0_-1", 0.9, (-1.9180814027786255, 0.0))
    d = torch.from_numpy(g["depth_norm"]).cuda()
    assert rel_l1(pp.area_resize_affine(d, (41, 55), [dA], [dB]).cpu().numpy(), g["depth_full"]) < 1e-6
    # bench-resolution shapes against the oracle: 320x427 -> 160x213 (the shipped config) and back
    rng = np.random.default_rng(3)
    big = rng.uniform(0, 255, (2, 3, 320, 427)).astype(np.float32)
    got = pp.area_resize_affine(torch.from_numpy(big).cuda(), (160, 213), [1 / 255.0], [0.0]).cpu().numpy()
    assert rel_l1(got, pr.normalize_tactile(pr.area_resize(big, (160, 213)), "0_255_to_0_1", 0.9)) < 1e-6
    small = rng.random((2, 1, 160, 213)).astype(np.float32)
    up = pp.sample_multi_channel_image_to_desired_size(torch.from_numpy(small).cuda(), (320, 427)).cpu().numpy()
    assert rel_l1(up, pr.area_resize(small, (320, 427))) < 1e-6


def test_predict_depth_pipeline_and_ema_eval_and_checkpoint(tmp_path):
    from gelslim_depth_amd import processing as pp
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    from oracle import processing_ref as pr
    from oracle import unet_numpy as on
    dims = [8, 16, 32]
    st = synth.make_state(3, 1, dims, 21, "conditioned")
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to("cuda").eval()
    cfg = types.SimpleNamespace(input_tactile_image_size=(40, 53), interp_method="area",
                                image_normalization_method="0_255_to_0_1", image_normalization_parameters=None,
                                depth_normalization_method="min_max_to_0_-1",
                                depth_normalization_parameters=(-1.9180814027786255, 0.0), norm_scale=0.9)
    rng = np.random.default_rng(5)
    img = rng.uniform(0, 255, (2, 3, 80, 107)).astype(np.float32)
    base = rng.uniform(0, 255, (2, 3, 80, 107)).astype(np.float32)
    got = pp.predict_depth_from_RGB(torch.from_numpy(img).cuda(), m, (80, 107), cfg,
                                    base_images=torch.from_numpy(base).cuda()).cpu().numpy()
    x = pr.normalize_tactile(pr.area_resize(pr.difference_image(img, base), (40, 53)), "0_255_to_0_1", 0.9)
    d = on.UNetOracle(st).forward(x, train=False)
    ref = pr.area_resize(pr.denormalize_depth(d, "min_max_to_0_-1", 0.9, (-1.9180814027786255, 0.0)), (80, 107))
    assert rel_l1(got, ref) < 1e-4

    # N2 / N3: train two steps, evaluate under the EMA weights without swapping, save, reload
    m.train()
    step = TrainStep(m)
    xb, tb = synth.make_batch(2, 40, 53, 22)
    xd, td = torch.from_numpy(xb).cuda(), torch.from_numpy(tb).cuda()
    step(xd, td)
    step(xd, td)
    y_ema = step.evaluate(xd, use_ema=True)
    y_live = step.evaluate(xd, use_ema=False)
    assert not torch.equal(y_ema, y_live)
    p_before = step.p_flat.clone()
    path = tmp_path / "unet_test.pth"
    step.save_checkpoint(str(path))
    assert torch.equal(step.p_flat, p_before)                   # nothing was swapped in place
    sd = torch.load(path, map_location="cpu")
    assert list(sd.keys()) == list(st.keys())
    m2 = UNet(n_channels=3, n_classes=1, layer_dimensions=dims)
    m2.load_state_dict(sd, strict=True)
    m2 = m2.to("cuda").eval()
    assert torch.equal(m2(x=xd), y_ema)                         # checkpoint == EMA weights + live BN buffers
    # oracle: same two steps, EMA shadow, eval forward
    net, _, _, shadow = on.train_steps(st, xb, tb, 2)
    st_ema = dict(net.s)
    st_ema.update(shadow)
    assert rel_l1(y_ema.cpu().numpy(), on.UNetOracle(st_ema).forward(xb, train=False)) < 1e-3


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graphed_inference_equals_eager_and_follows_weight_updates(precision):
    """hipGraph replay of the eval forward: bit-identical to the eager schedule, and it re-reads the parameters."""
    from gelslim_depth_amd.graph import GraphedInference
    from gelslim_depth_amd.models.unet import UNet
    dims = [32, 64, 128]
    st = synth.make_state(3, 1, dims, 31, "conditioned")
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision=precision)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to("cuda").eval()
    rng = np.random.default_rng(2)
    x1 = torch.from_numpy(rng.random((2, 3, 40, 53)).astype(np.float32)).cuda()
    x2 = torch.from_numpy(rng.random((2, 3, 40, 53)).astype(np.float32)).cuda()
    fn = GraphedInference(m, x1)
    with torch.no_grad():
        for x in (x1, x2, x1):
            assert torch.equal(fn(x), m(x=x))
        for p in m.parameters():                      # in-place update, as an optimiser step does
            p.mul_(1.01)
        assert torch.equal(fn(x2), m(x=x2))
    with pytest.raises(RuntimeError):
        fn(torch.zeros((1, 3, 40, 53), device="cuda"))
