"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's dataset path:
/root/reference/gelslim_depth/datasets/general_dataset.py (GeneralDataset) on in-memory object dicts, with torch CPU
operators where the reference uses them (torch.cat, torch.randperm, F.interpolate(mode='area'), .min/.max/.mean/.std).

GeneralDataset itself cannot be imported in the build container (it imports torchvision, which is absent: an ordinary
ModuleNotFoundError), so the class glue is restated here from its text; the normalisers it calls ARE importable and
tests/golden/make_golden.py::g_dataset runs this restatement WITH the reference's own normalize_tactile_image /
normalize_depth_image injected to produce tests/golden/gdataset.npz.  tests/test_oracle.py::test_dataset_oracle pins the
default (self-contained) normalisers below against that file.
"""
import numpy as np
import torch
import torch.nn.functional as F


def normalize_tactile(x, method, norm_scale, params=None):          # normalization_utils.py:4-35
    if "0_255" not in method:
        mins, maxes, means, stds = params
    if method == "mean_std":
        scale, bias, den = 1.0, means, stds
    elif method == "0_255_to_-1_1":
        scale, bias, den = 2.0, [127.5], [255.0]
    elif method == "0_255_to_0_1":
        scale, bias, den = 1.0, [0.0], [255.0]
    else:       # 'min_max_to_-1_1' raises TypeError inside the reference (list * float, normalization_utils.py:9)
        raise TypeError(method)
    out = torch.zeros_like(x)
    cdim = 0 if x.dim() == 3 else 1
    for i in range(x.shape[cdim]):
        sl = (i,) if cdim == 0 else (slice(None), i)
        out[sl] = scale * (x[sl] - bias[min(i, len(bias) - 1)]) / den[min(i, len(den) - 1)]
    return out


def normalize_depth(d, method, norm_scale, params=None):            # normalization_utils.py:67-99
    vals = list(params) if params is not None else []
    mn, mx, mean, std = (vals + [None] * 4)[:4]
    if method == "min_max_to_-1_1":
        scale, bias, den = norm_scale, 0.5 * (mx + mn), mx - mn
    elif method == "mean_std":
        scale, bias, den = 1.0, mean, std
    elif method == "min_max_to_0_1":
        scale, bias, den = norm_scale, mn, mx - mn
    elif method == "min_max_to_0_-1":
        scale, bias, den = -norm_scale, mn, mx - mn
    return scale * (d - bias) / den


def difference_image(a, b):                                          # image_utils.py:6-10
    return (a - b + 255.0) / 2.0


def resize(x, size):                                                 # image_utils.py:12-15
    return F.interpolate(x, size=size, mode="area")


def gaussian_blur(x, kernel_size):                                   # image_utils.py:17-19 -> torchvision TF.gaussian_blur
    """torchvision.transforms.functional.gaussian_blur(x, kernel_size) restated with torch CPU operators.  torchvision is a
    third-party dependency that the reference does not pin (setup.py / requirements.txt do not list it) and that is absent
    from this image, so this follows its published algorithm (functional.gaussian_blur -> _functional_tensor.gaussian_blur):
    sigma = 0.15 k + 0.35; 1-D kernel exp(-0.5 (x/sigma)^2) on linspace(-(k-1)/2, (k-1)/2, k), normalised; 2-D kernel =
    outer product; reflect padding of k//2 on every side; depthwise conv2d.  PARITY UNPINNED for this one function."""
    k = int(kernel_size)
    sigma = k * 0.15 + 0.35
    half = (k - 1) * 0.5
    pts = torch.linspace(-half, half, steps=k, dtype=x.dtype)
    pdf = torch.exp(-0.5 * (pts / sigma).pow(2))
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :])
    c = x.shape[-3]
    w = k2.expand(c, 1, k, k)
    xp = F.pad(x, [k // 2, k // 2, k // 2, k // 2], mode="reflect")
    return F.conv2d(xp, w, groups=c)


class DatasetOracle:
    """GeneralDataset on in-memory dicts (general_dataset.py:12-245), sequential loading branch."""

    def __init__(self, objects, extra_objects=None, use_difference_image=False,
                 depth_normalization_method="min_max_to_0_-1", image_normalization_method="mean_std",
                 separate_fingers=True, downsample_factor=0.5, depth_normalization_parameters=None,
                 image_normalization_parameters=None, norm_scale=None, max_datapoints_per_object=None,
                 normalizers=None, depth_image_blur_kernel=1):
        self.depth_image_blur_kernel = depth_image_blur_kernel
        self.use_difference_image = use_difference_image
        self.downsample_factor = downsample_factor
        self.max_datapoints_per_object = max_datapoints_per_object
        self.separate_fingers = separate_fingers
        self.input_tactile_image_size = None
        self.norm_tactile, self.norm_depth = normalizers or (normalize_tactile, normalize_depth)
        ent = {}
        for i, obj in enumerate(objects):                            # :173-180
            for k, v in self._load(obj, i, True).items():
                ent[k] = torch.cat((ent[k], v), dim=0) if k in ent else v
        for i, obj in enumerate(extra_objects or []):                # :181-190
            for k, v in self._load(obj, i, False).items():
                ent[k] = torch.cat((ent[k], v), dim=0) if k in ent else v
        self.entire_dataset = ent
        self.depth_normalization_method = depth_normalization_method
        self.image_normalization_method = image_normalization_method
        t = ent["tactile_image"]
        self.input_tactile_image_size = (t.shape[2], t.shape[3])     # :48
        if depth_normalization_parameters is None:                   # :49-52, 199-204
            d = ent["depth_image"]
            depth_normalization_parameters = (d.min().item(), d.max().item(), d.mean().item(), d.std().item())
        self.depth_normalization_parameters = depth_normalization_parameters
        if image_normalization_parameters is None:                   # :53-56, 206-220
            mins, maxes, means, stds = [], [], [], []
            for c in range(t.shape[1]):
                ch = t[:, c, ...]
                maxes.append(ch.max().item()); mins.append(ch.min().item())          # noqa: E702
                means.append(ch.mean().item()); stds.append(ch.std().item())         # noqa: E702
            image_normalization_parameters = (mins, maxes, means, stds)
        self.image_normalization_parameters = image_normalization_parameters
        self.norm_scale = norm_scale

    def _load(self, obj, object_index, main):                        # :60-97 (main) / :99-134 (extra)
        tac, dep = obj["tactile_image"].float(), obj["depth_image"].float()
        if main and self.input_tactile_image_size is None:           # :65-66
            self.input_tactile_image_size = (int(tac.shape[2] * self.downsample_factor),
                                             int(tac.shape[3] * self.downsample_factor))
        size = self.input_tactile_image_size
        if self.use_difference_image:
            tac = difference_image(tac, obj["base_tactile_image"].float())
        if self.separate_fingers:                                    # :68-76
            tc, dc = tac.shape[1] // 2, dep.shape[1] // 2
            tac = torch.cat((tac[:, 0:tc], tac[:, tc:2 * tc]), dim=0)
            dep = torch.cat((dep[:, 0:dc], dep[:, dc:2 * dc]), dim=0)
        data = {"tactile_image": resize(tac, size), "depth_image": resize(dep, size)}
        if self.depth_image_blur_kernel > 1:                         # :74-76, 84-86
            data["depth_image"] = gaussian_blur(data["depth_image"], self.depth_image_blur_kernel)
        rows = data["tactile_image"].shape[0]
        data["object_index"] = torch.tensor([object_index] * rows)   # :87
        if self.max_datapoints_per_object is not None and rows > self.max_datapoints_per_object:     # :90-96
            idx = torch.randperm(rows)[: self.max_datapoints_per_object]
            data = {k: v[idx, ...] for k, v in data.items()}
        return data

    def __len__(self):
        return self.entire_dataset["tactile_image"].shape[0]

    def __getitem__(self, idx):                                      # :222-245
        e = self.entire_dataset
        return {"tactile_image": self.norm_tactile(e["tactile_image"][idx, ...], self.image_normalization_method,
                                                   self.norm_scale, self.image_normalization_parameters),
                "depth_image": self.norm_depth(e["depth_image"][idx, ...], self.depth_normalization_method,
                                               self.norm_scale, self.depth_normalization_parameters),
                "object_index": e["object_index"][idx]}


def synthetic_objects(seed, counts, h=20, w=26):
    """Deterministic stand-ins for the reference's per-object .pt files: K x 6 x h x w tactile + base images with
    integer values 0..255 (camera bytes as float) and K x 2 x h x w depth in [-2, 0]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    objs = []
    for k in counts:
        objs.append({"tactile_image": torch.from_numpy(rng.integers(0, 256, (k, 6, h, w)).astype(np.float32)),
                     "base_tactile_image": torch.from_numpy(rng.integers(0, 256, (k, 6, h, w)).astype(np.float32)),
                     "depth_image": torch.from_numpy((-2.0 * rng.random((k, 2, h, w))).astype(np.float32))})
    return objs


def loader_order(n, batch_size):
    """Sample order of DataLoader(dataset, batch_size, shuffle=True) under the current torch RNG state
    (train_unet.py:229), taken from torch's own DataLoader."""
    from torch.utils.data import DataLoader, TensorDataset
    return [b[0] for b in DataLoader(TensorDataset(torch.arange(n)), batch_size=batch_size, shuffle=True)]
