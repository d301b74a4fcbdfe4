"""GPU parity for SURVEY 8(f) N4: the device-resident dataset path (gelslim_depth_amd/dataset.py, gsd_dataset.hip)
against the golden fixture made with the reference's normalisers and against oracle/dataset_ref.py."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from gelslim_depth_amd import synth

pytestmark = pytest.mark.gpu

CASES = (("a", dict(use_difference_image=True, image_normalization_method="0_255_to_0_1",
                    depth_normalization_method="min_max_to_0_-1", norm_scale=0.9, max_datapoints_per_object=5)),
         ("b", dict(use_difference_image=False, image_normalization_method="mean_std",
                    depth_normalization_method="mean_std", norm_scale=1.0, separate_fingers=False)))


@pytest.mark.parametrize("tag,kw", CASES)
def test_device_dataset_vs_golden(tag, kw):
    from gelslim_depth_amd.dataset import DeviceDataset, DeviceLoader
    from oracle import dataset_ref as dr
    g = load_golden("gdataset.npz")
    torch.manual_seed(7)
    ds = DeviceDataset(objects=dr.synthetic_objects(11, [3, 4]), extra_objects=dr.synthetic_objects(12, [2]),
                       device="cuda", interp_method="area", **kw)
    assert len(ds) == g[tag + "_tactile"].shape[0]
    assert ds.input_tactile_image_size == (10, 13)
    assert np.array_equal(ds.entire_dataset["object_index"].cpu().numpy(), g[tag + "_object_index"])
    # integer-valued inputs, windows of 4 pixels: the resized set is exact in fp32 up to the rounding order of the mean
    assert np.abs(ds.entire_dataset["tactile_image"].cpu().numpy() - g[tag + "_tactile_raw"]).max() <= 2e-5
    assert np.abs(ds.entire_dataset["depth_image"].cpu().numpy() - g[tag + "_depth_raw"]).max() <= 1e-6
    # statistics: fp64 two-stage reduction here, torch's fp32 reductions in the reference => 1e-6 relative
    assert np.allclose(np.array(ds.depth_normalization_parameters), g[tag + "_depth_params"], rtol=2e-6, atol=1e-6)
    assert np.allclose(np.array(ds.image_normalization_parameters), g[tag + "_image_params"], rtol=2e-6, atol=1e-5)
    tac = np.stack([ds[i]["tactile_image"].cpu().numpy() for i in range(len(ds))])
    dep = np.stack([ds[i]["depth_image"].cpu().numpy() for i in range(len(ds))])
    assert np.abs(tac - g[tag + "_tactile"]).max() <= 2e-6
    assert np.abs(dep - g[tag + "_depth"]).max() <= 2e-6
    # the loader: same order as torch's DataLoader under the same seed, ragged last batch kept, batches == samples
    torch.manual_seed(21)
    batches = list(DeviceLoader(ds, batch_size=4, shuffle=True))
    order = np.concatenate([np.arange(0)] + [g[tag + "_order"]])
    assert [b["tactile_image"].shape[0] for b in batches] == [4] * (len(ds) // 4) + ([len(ds) % 4] if len(ds) % 4 else [])
    got = np.concatenate([b["tactile_image"].cpu().numpy() for b in batches])
    assert np.array_equal(got, tac[order])
    gotd = np.concatenate([b["depth_image"].cpu().numpy() for b in batches])
    assert np.array_equal(gotd, dep[order])
    assert np.array_equal(np.concatenate([b["object_index"].cpu().numpy() for b in batches]), g[tag + "_object_index"][order])
    # two ranks split every global batch contiguously and together cover it; both yield the same number of batches of the
    # same sizes (the ragged last global batch is padded by wrapping around to the start of the permutation)
    torch.manual_seed(21)
    r0 = list(DeviceLoader(ds, batch_size=2, shuffle=True, rank=0, world_size=2))
    torch.manual_seed(21)
    r1 = list(DeviceLoader(ds, batch_size=2, shuffle=True, rank=1, world_size=2))
    assert len(r0) == len(r1) == len(DeviceLoader(ds, batch_size=2, rank=0, world_size=2))
    assert [b["depth_image"].shape[0] for b in r0] == [b["depth_image"].shape[0] for b in r1]
    both = []
    for a, b in zip(r0, r1):
        both += [a["depth_image"].cpu().numpy(), b["depth_image"].cpu().numpy()]
    both = np.concatenate(both)
    padded = np.concatenate([order, order[:both.shape[0] - len(order)]])
    assert np.array_equal(both, dep[padded])


def test_device_dataset_uint8_and_oracle_at_reference_size():
    """uint8 camera frames at the reference's raw size (320x427 -> downsample 0.5 -> 160x213), checked against the oracle;
    plus gather_affine's non-vectorised path (HW odd) and the stats kernel on a larger tensor."""
    from gelslim_depth_amd.dataset import DeviceDataset, channel_stats
    from oracle import dataset_ref as dr
    rng = np.random.default_rng(5)
    objs = [{"tactile_image": torch.from_numpy(rng.integers(0, 256, (2, 6, 320, 427), dtype=np.uint8)),
             "base_tactile_image": torch.from_numpy(rng.integers(0, 256, (2, 6, 320, 427), dtype=np.uint8)),
             "depth_image": torch.from_numpy((-2 * rng.random((2, 2, 320, 427))).astype(np.float32))}]
    kw = dict(use_difference_image=True, image_normalization_method="0_255_to_-1_1",
              depth_normalization_method="min_max_to_0_1", norm_scale=0.9)
    ds = DeviceDataset(objects=objs, device="cuda", **kw)
    ref = dr.DatasetOracle(objs, **kw)
    assert ds.input_tactile_image_size == ref.input_tactile_image_size == (160, 213)
    assert np.abs(ds.entire_dataset["tactile_image"].cpu().numpy() - ref.entire_dataset["tactile_image"].numpy()).max() <= 3e-5
    assert np.allclose(np.array(ds.depth_normalization_parameters), np.array(ref.depth_normalization_parameters),
                       rtol=2e-6, atol=1e-6)
    for i in (0, 3):
        a, b = ds[i], ref[i]
        assert np.abs(a["tactile_image"].cpu().numpy() - b["tactile_image"].numpy()).max() <= 2e-6
        assert np.abs(a["depth_image"].cpu().numpy() - b["depth_image"].numpy()).max() <= 2e-6
    x = torch.from_numpy(rng.normal(3.0, 2.0, (7, 3, 33, 31)).astype(np.float32))
    s = channel_stats(x.cuda()).cpu().numpy()
    for c in range(3):
        ch = x[:, c].double()
        assert np.allclose(s[c], [ch.min().item(), ch.max().item(), ch.mean().item(), ch.std().item()], rtol=1e-9, atol=1e-9)
    with pytest.raises(IndexError):
        ds.batch(torch.tensor([0, len(ds)]))


def test_train_epoch_from_device_dataset():
    """End to end: DeviceDataset -> DeviceLoader -> TrainStep for one epoch equals the oracle trainer fed the oracle
    dataset's batches in the same order (train_unet.py:340-377)."""
    from gelslim_depth_amd.dataset import DeviceDataset, DeviceLoader, train_epoch
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    from oracle import dataset_ref as dr
    from oracle import torch_cpu_path as ot
    kw = dict(use_difference_image=True, image_normalization_method="0_255_to_0_1",
              depth_normalization_method="min_max_to_0_-1", norm_scale=0.9)
    objs = dr.synthetic_objects(31, [3, 2], h=42, w=54)
    dims = [4, 8, 16]
    st = synth.make_state(3, 1, dims, 9, "conditioned")
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to("cuda").train()
    step = TrainStep(m, lr=1e-3, weight_decay=1e-6, ema_decay=0.995, loss="mse")
    ds = DeviceDataset(objects=objs, device="cuda", **kw)
    torch.manual_seed(3)
    total, nb = train_epoch(step, DeviceLoader(ds, batch_size=4, shuffle=True))
    ref = dr.DatasetOracle(objs, **kw)
    torch.manual_seed(3)
    tr = ot.CpuTrainer(st)
    ref_total = 0.0
    for idx in dr.loader_order(len(ref), 4):
        xs = torch.stack([ref[int(i)]["tactile_image"] for i in idx])
        ts = torch.stack([ref[int(i)]["depth_image"] for i in idx])
        ref_total += float(tr.step(xs, ts))
    assert nb == 3
    assert abs(total - ref_total) <= 1e-3 * abs(ref_total)


def test_fit_two_epochs_end_to_end(tmp_path):
    """harness.fit on the device-resident dataset: train pass, EMA-weights validation/test passes, best-validation
    checkpoint in the reference layout, reference log lines; the validation loss equals the oracle's evaluation of the
    saved checkpoint on the same batches."""
    from gelslim_depth_amd import harness
    from gelslim_depth_amd.dataset import DeviceDataset, DeviceLoader
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    from oracle import dataset_ref as dr
    from oracle import torch_cpu_path as ot
    kw = dict(use_difference_image=True, image_normalization_method="0_255_to_0_1",
              depth_normalization_method="min_max_to_0_-1", norm_scale=0.9)
    dims = [8, 16, 32]
    st = synth.make_state(3, 1, dims, 4, "conditioned")
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to("cuda").train()
    step = TrainStep(m, lr=1e-3, weight_decay=1e-6, ema_decay=0.995, loss="mse")
    train_ds = DeviceDataset(objects=dr.synthetic_objects(41, [3, 3], h=42, w=54), device="cuda", **kw)
    val_objs = dr.synthetic_objects(42, [2], h=42, w=54)
    val_ds = DeviceDataset(objects=val_objs, device="cuda", depth_normalization_parameters=train_ds.depth_normalization_parameters, **kw)
    lines = []
    torch.manual_seed(0)
    H = harness.fit(step, DeviceLoader(train_ds, 4, shuffle=True), DeviceLoader(val_ds, 4), DeviceLoader(val_ds, 2),
                    str(tmp_path / "weights"), "unet_t", loss_values_path=str(tmp_path / "loss.txt"), train_indefinitely=True,
                    max_epochs=2, echo=lines.append)
    assert len(H["train_loss"]) == 2 and all(np.isfinite(v) for k in H for v in H[k])
    assert lines[0] == "Validation loss is at a minimum. Saving the model" and lines[1] == "[INFO] EPOCH: 1"
    assert lines[2] == "Train loss: {:.6f},  Validation loss: {:.6f}, Test loss: {:.6f}".format(
        H["train_loss"][0], H["validation_loss"][0], H["test_loss"][0])
    assert lines[-2] == "Training complete"
    # the checkpoint is the EMA-swapped reference layout: evaluate it with the oracle on the validation set
    sd = torch.load(tmp_path / "weights" / "unet_t.pth")
    assert len(sd) == len(m.state_dict())
    ref_ds = dr.DatasetOracle(val_objs, depth_normalization_parameters=train_ds.depth_normalization_parameters, **kw)
    xs = torch.stack([ref_ds[i]["tactile_image"] for i in range(len(ref_ds))])
    ts = torch.stack([ref_ds[i]["depth_image"] for i in range(len(ref_ds))])
    with torch.no_grad():
        out = ot.forward({k: v.clone() for k, v in sd.items()}, xs, train=False)
    best = int(np.argmin(H["validation_loss"]))
    assert abs(float(((out - ts) ** 2).mean()) - H["validation_loss"][best]) <= 1e-4 * H["validation_loss"][best]


@pytest.mark.parametrize("n,c,h,w,k", [(2, 1, 10, 13, 3), (3, 2, 17, 9, 5), (1, 1, 160, 213, 9), (2, 1, 6, 7, 11), (1, 1, 4, 4, 1)])
def test_gaussian_blur_vs_oracle(n, c, h, w, k):
    """gsd_gaussian_blur == oracle/dataset_ref.gaussian_blur (torchvision's published algorithm on torch CPU operators:
    reflect padding + depthwise conv2d with the outer-product kernel), the padding as wide as reflect allows (k//2 = H - 1)."""
    from gelslim_depth_amd.dataset import gaussian_blur, gaussian_kernel2d
    from gelslim_depth_amd import _lib as gsd
    from oracle import dataset_ref as dr
    rng = np.random.default_rng(h * w + k)
    x = torch.from_numpy((-2 * rng.random((n, c, h, w))).astype(np.float32))
    ref = dr.gaussian_blur(x, k).numpy()
    xd = x.cuda()
    got = gaussian_blur(xd, k).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max()
    k2 = gaussian_kernel2d(k)
    assert abs(float(k2.sum()) - 1.0) < 1e-6 and torch.equal(k2, k2.t())
    # refused, not mis-read: even sizes, in-place, a pad that reflect cannot serve
    k2d, out = k2.cuda(), torch.empty_like(xd)
    assert gsd.lib.gsd_gaussian_blur(xd.data_ptr(), n * c, h, w, k2d.data_ptr(), 4, out.data_ptr(), gsd.stream_ptr()) == -1
    assert gsd.lib.gsd_gaussian_blur(xd.data_ptr(), n * c, h, w, k2d.data_ptr(), k, xd.data_ptr(), gsd.stream_ptr()) == -1
    big = 2 * min(h, w) + 1
    assert gsd.lib.gsd_gaussian_blur(xd.data_ptr(), n * c, h, w, k2d.data_ptr(), big, out.data_ptr(), gsd.stream_ptr()) == -2


@pytest.mark.parametrize("separate", [True, False])
def test_device_dataset_with_depth_blur(separate):
    """depth_image_blur_kernel > 1 (general_dataset.py:74-76,84-86): the depth targets are blurred after the area resize,
    before the per-object subsample and the normalisation statistics; the tactile images are not."""
    from gelslim_depth_amd.dataset import DeviceDataset
    from oracle import dataset_ref as dr
    kw = dict(use_difference_image=True, image_normalization_method="0_255_to_0_1", depth_normalization_method="min_max_to_0_-1",
              norm_scale=0.9, max_datapoints_per_object=5, separate_fingers=separate)
    torch.manual_seed(7)
    ds = DeviceDataset(objects=dr.synthetic_objects(11, [3, 4]), extra_objects=dr.synthetic_objects(12, [2]), device="cuda",
                       depth_image_blur_kernel=5, **kw)
    torch.manual_seed(7)
    ref = dr.DatasetOracle(dr.synthetic_objects(11, [3, 4]), dr.synthetic_objects(12, [2]), depth_image_blur_kernel=5, **kw)
    torch.manual_seed(7)
    plain = dr.DatasetOracle(dr.synthetic_objects(11, [3, 4]), dr.synthetic_objects(12, [2]), **kw)
    d, dref = ds.entire_dataset["depth_image"].cpu().numpy(), ref.entire_dataset["depth_image"].numpy()
    assert d.shape == dref.shape and np.abs(d - dref).max() <= 2e-6 * np.abs(dref).max()
    assert np.abs(dref - plain.entire_dataset["depth_image"].numpy()).max() > 1e-3          # the blur did something
    assert np.abs(ds.entire_dataset["tactile_image"].cpu().numpy() - ref.entire_dataset["tactile_image"].numpy()).max() <= 2e-5
    assert np.allclose(np.array(ds.depth_normalization_parameters), np.array(ref.depth_normalization_parameters), rtol=2e-6, atol=1e-6)
    for i in (0, len(ds) - 1):
        assert np.abs(ds[i]["depth_image"].cpu().numpy() - ref[i]["depth_image"].numpy()).max() <= 3e-6
    with pytest.raises(ValueError):
        DeviceDataset(objects=dr.synthetic_objects(11, [1]), device="cuda", depth_image_blur_kernel=4)
