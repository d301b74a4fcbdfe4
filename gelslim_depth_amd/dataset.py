"""Device-resident dataset path: the reference's GeneralDataset + DataLoader pair with the
tensors kept in HBM and every per-pixel operation a libgsd kernel.

Mirrors (paths under /root/reference/):
  GeneralDataset.__init__ keywords / attributes   gelslim_depth/datasets/general_dataset.py:12-58
  load_object_dataset / load_extra_object_dataset general_dataset.py:60-97, 99-134   (finger split, difference image,
                                                  area resize, per-object subsample through torch.randperm)
  load_entire_dataset                             general_dataset.py:136-192  (sequential order: main list, then extra)
  calculate_*_normalization_params                general_dataset.py:199-220
  normalize_sample / __getitem__                  general_dataset.py:222-245
  DataLoader(shuffle=True) around it              train_utils/train_unet.py:229-233, consumed at :340-347

What changes against the reference: raw object tensors are uploaded once, one kernel does split + difference + resize
straight into the dataset arena (no torch.cat growth, no host copy of the resized set), statistics are one reduction per
tensor, and a batch is ONE gather+normalise kernel per tensor driven by a device index vector -- no per-sample Python
__getitem__, no collate, no pinned H2D in the step.  The shuffle order is drawn exactly as torch's RandomSampler draws
it, so with the same torch seed the batches are the ones the reference's DataLoader would produce.
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from ._lib import lib, check
from .processing import tactile_affine


def depth_norm_affine(method: str, norm_scale: float, params=None) -> Tuple[float, float]:
    """(A, B) with normalize_depth_image(d) == A*d + B  (normalization_utils.py:67-99)."""
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
    else:
        raise ValueError(f"unknown depth_normalization_method {method!r}")
    return scale / den, -scale * bias / den


def _need_cuda(device) -> torch.device:
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise L.GsdError("DeviceDataset keeps the dataset in HBM and has no CPU path; pass a cuda device")
    return dev


def ingest_images(raw: torch.Tensor, base: Optional[torch.Tensor], c0: int, c1: int, size: Tuple[int, int],
                  out: torch.Tensor) -> None:
    """out (K, c1-c0, OH, OW) <- area_resize(diff(raw[:, c0:c1], base[:, c0:c1])) -- one kernel, channel view by stride."""
    if raw.dtype not in (torch.float32, torch.uint8):
        raw = raw.float()
    if base is not None and base.dtype != raw.dtype:
        raw, base = raw.float(), base.float()
    raw = raw.contiguous()
    k, c, h, w = raw.shape
    bptr, bns, bcs = None, 0, 0
    if base is not None:
        base = base.expand_as(raw).contiguous()
        bptr, bns, bcs = base.data_ptr() + c0 * h * w * base.element_size(), c * h * w, h * w
    if not (out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (k, c1 - c0, size[0], size[1])):
        raise L.GsdError("ingest_images: bad output tensor")
    for s in range(0, k, 32768):        # grid.z limit
        e = min(k, s + 32768)
        check(lib.gsd_ingest_images(raw.data_ptr() + (s * c + c0) * h * w * raw.element_size(),
                                    None if bptr is None else bptr + s * c * h * w * base.element_size(),
                                    0 if raw.dtype == torch.float32 else 1, e - s, c1 - c0, h, w, c * h * w, h * w, bns, bcs,
                                    out[s:e].data_ptr(), size[0], size[1], 255.0, 0.5, L.stream_ptr()), "ingest_images")


def gaussian_kernel2d(kernel_size: int, sigma: Optional[float] = None) -> torch.Tensor:
    """The K x K kernel torchvision.transforms.functional.gaussian_blur(img, kernel_size) convolves with (fp32, CPU):
    sigma = 0.3 * ((K - 1) * 0.5 - 1) + 0.8 unless given; k1[i] = exp(-0.5 (x_i / sigma)^2) on K points of
    linspace(-(K-1)/2, (K-1)/2), normalised to sum 1; k2 = k1 (column) x k1 (row).  torchvision is absent from this image and
    unpinned by the reference: restated from its published source, parity unpinned."""
    k = int(kernel_size)
    if k < 1 or k % 2 == 0:
        raise ValueError(f"Kernel size value should be an odd and positive number, got {kernel_size}")
    if sigma is None:
        sigma = k * 0.15 + 0.35            # == 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    half = (k - 1) * 0.5
    x = torch.linspace(-half, half, steps=k, dtype=torch.float32)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    k1 = pdf / pdf.sum()
    return torch.mm(k1[:, None], k1[None, :])


def gaussian_blur(x: torch.Tensor, kernel_size: int) -> torch.Tensor:
    """blur_depth_images (image_utils.py:17-19) on a contiguous (N, C, H, W) fp32 device tensor: one libgsd kernel."""
    if x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous() or x.dim() != 4:
        raise L.GsdError("gaussian_blur: expected a contiguous (N,C,H,W) float32 tensor on the GPU")
    k2 = gaussian_kernel2d(kernel_size).to(x.device)
    out = torch.empty_like(x)
    n, c, h, w = x.shape
    if n * c:
        check(lib.gsd_gaussian_blur(x.data_ptr(), n * c, h, w, k2.data_ptr(), int(kernel_size), out.data_ptr(), L.stream_ptr()),
              "gaussian_blur")
    return out


def channel_stats(x: torch.Tensor) -> torch.Tensor:
    """(C, 4) float64 {min, max, mean, unbiased std} per channel of x (N, C, H, W)."""
    n, c, h, w = x.shape
    out = torch.empty((c, 4), device=x.device, dtype=torch.float64)
    ws = torch.empty((int(lib.gsd_channel_stats_workspace(c)),), device=x.device, dtype=torch.float64)
    check(lib.gsd_channel_stats(x.data_ptr(), n, c, h * w, out.data_ptr(), ws.data_ptr(), L.stream_ptr()), "channel_stats")
    return out


def gather_affine(src: torch.Tensor, idx: torch.Tensor, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    m, c, h, w = src.shape
    out = torch.empty((idx.numel(), c, h, w), device=src.device, dtype=torch.float32)
    check(lib.gsd_gather_affine(src.data_ptr(), idx.data_ptr(), m, idx.numel(), c, h * w, A.data_ptr(), B.data_ptr(),
                                A.numel(), out.data_ptr(), L.stream_ptr()), "gather_affine")
    return out


class DeviceDataset:
    """Drop-in for GeneralDataset (general_dataset.py:12): same keywords, same attributes
    (`entire_dataset`, `input_tactile_image_size`, `depth_normalization_parameters`, `image_normalization_parameters`,
    `norm_scale`), same `len()` / `[idx]` results -- held on `device`.

    `objects` / `extra_objects` (lists of already loaded {'tactile_image','base_tactile_image','depth_image'} dicts) may
    replace `directory`+`pt_file_list` / `extra_directory`+`extra_pt_list` for in-memory or synthetic data."""

    def __init__(self, directory=None, pt_file_list=None, extra_directory=None, extra_pt_list=None,
                 use_difference_image=False, depth_normalization_method="min_max_to_0_-1",
                 image_normalization_method="mean_std", separate_fingers=True, downsample_factor=0.5,
                 depth_image_blur_kernel: int = 1, depth_normalization_parameters=None, image_normalization_parameters=None,
                 norm_scale=None, max_datapoints_per_object=None, device=None, interp_method=None, objects=None,
                 extra_objects=None) -> None:
        if objects is None:
            assert directory is not None and os.path.exists(directory), f"Dataset path {directory} does not exist"
        if interp_method not in (None, "area"):
            raise NotImplementedError("only interp_method='area' (what the reference's configs use) is implemented")
        if depth_image_blur_kernel > 1:
            gaussian_kernel2d(depth_image_blur_kernel)      # an even size raises here, as torchvision does
        self.use_difference_image = use_difference_image
        self.downsample_factor = downsample_factor
        self.depth_image_blur_kernel = depth_image_blur_kernel
        self.dataset_path = directory
        self.pt_file_list = pt_file_list
        self.extra_directory = extra_directory
        self.extra_pt_list = extra_pt_list
        self.max_datapoints_per_object = max_datapoints_per_object
        self.separate_fingers = separate_fingers
        self.device = _need_cuda(device)
        self.interp_method = interp_method or "area"
        self.input_tactile_image_size = None
        self._objects, self._extra_objects = objects, extra_objects
        self.entire_dataset = self.load_entire_dataset()
        self.depth_normalization_method = depth_normalization_method
        self.image_normalization_method = image_normalization_method
        t = self.entire_dataset["tactile_image"]
        self.input_tactile_image_size = (t.shape[2], t.shape[3])
        self.depth_normalization_parameters = (self.calculate_depth_normalization_params()
                                               if depth_normalization_parameters is None else depth_normalization_parameters)
        self.image_normalization_parameters = (self.calculate_image_normalization_params()
                                               if image_normalization_parameters is None else image_normalization_parameters)
        self.norm_scale = norm_scale
        self._affine_cache = None

    # ---- loading ------------------------------------------------------------------------------------------------
    def _read(self, extra: bool, object_index: int) -> Dict[str, torch.Tensor]:
        mem = self._extra_objects if extra else self._objects
        if mem is not None:
            return mem[object_index]
        folder, names = (self.extra_directory, self.extra_pt_list) if extra else (self.dataset_path, self.pt_file_list)
        return torch.load(os.path.join(folder, names[object_index]), map_location="cpu")

    def _load_object(self, extra: bool, object_index: int) -> Dict[str, torch.Tensor]:
        """general_dataset.py:60-97 (main) / 99-134 (extra; identical except the size is never (re)derived)."""
        data = self._read(extra, object_index)
        tac, dep = data["tactile_image"], data["depth_image"]
        base = data["base_tactile_image"] if self.use_difference_image else None
        k = tac.shape[0]
        if self.input_tactile_image_size is None:
            if extra:
                raise L.GsdError("extra objects need a main object first (the reference derives the size from it)")
            self.input_tactile_image_size = (int(tac.shape[2] * self.downsample_factor),
                                             int(tac.shape[3] * self.downsample_factor))
        size = self.input_tactile_image_size
        rows = 2 * k if self.separate_fingers else k
        # the reference keeps a random subset per object; drawn the same way here (same torch.randperm call on the
        # global CPU generator, one call per oversized object, in load order) so the same samples survive
        keep = None
        if self.max_datapoints_per_object is not None and rows > self.max_datapoints_per_object:
            keep = torch.randperm(rows)[: self.max_datapoints_per_object]
        dev = self.device
        tac_d, dep_d = tac.to(dev, non_blocking=True), dep.to(dev, non_blocking=True)
        base_d = base.to(dev, non_blocking=True) if base is not None else None
        if self.separate_fingers:
            tc, dc = tac.shape[1] // 2, dep.shape[1] // 2
            t_out = torch.empty((2 * k, tc, *size), device=dev, dtype=torch.float32)
            d_out = torch.empty((2 * k, dc, *size), device=dev, dtype=torch.float32)
            for f in range(2):      # torch.cat((x[:, 0:3], x[:, 3:6]), dim=0): all left fingers, then all right fingers
                ingest_images(tac_d, base_d, f * tc, (f + 1) * tc, size, t_out[f * k:(f + 1) * k])
                ingest_images(dep_d, None, f * dc, (f + 1) * dc, size, d_out[f * k:(f + 1) * k])
        else:
            t_out = torch.empty((k, tac.shape[1], *size), device=dev, dtype=torch.float32)
            d_out = torch.empty((k, dep.shape[1], *size), device=dev, dtype=torch.float32)
            ingest_images(tac_d, base_d, 0, tac.shape[1], size, t_out)
            ingest_images(dep_d, None, 0, dep.shape[1], size, d_out)
        if self.depth_image_blur_kernel > 1:        # general_dataset.py:74-76,84-86: blur AFTER the area resize
            d_out = gaussian_blur(d_out, self.depth_image_blur_kernel)
        obj = torch.full((rows,), object_index, dtype=torch.int64, device=dev)
        if keep is not None:
            kd = keep.to(dev)
            t_out, d_out, obj = t_out[kd], d_out[kd], obj[kd]
        return {"tactile_image": t_out, "depth_image": d_out, "object_index": obj}

    def load_entire_dataset(self) -> Dict[str, torch.Tensor]:
        """general_dataset.py:136-192, sequential branch: main objects in list order, then the extra objects."""
        parts: List[Dict[str, torch.Tensor]] = []
        n_main = len(self._objects) if self._objects is not None else len(self.pt_file_list)
        for i in range(n_main):
            parts.append(self._load_object(False, i))
        has_extra = self._extra_objects is not None or self.extra_directory is not None
        if has_extra:
            n_extra = len(self._extra_objects) if self._extra_objects is not None else len(self.extra_pt_list)
            for i in range(n_extra):
                parts.append(self._load_object(True, i))
        return {key: torch.cat([p[key] for p in parts], dim=0) for key in ("tactile_image", "depth_image", "object_index")}

    # ---- statistics ---------------------------------------------------------------------------------------------
    def calculate_depth_normalization_params(self):
        """(min, max, mean, std) over ALL depth values (general_dataset.py:199-204)."""
        d = self.entire_dataset["depth_image"]
        s = channel_stats(d.reshape(1, 1, -1, 1)).cpu()[0].tolist()
        return (s[0], s[1], s[2], s[3])

    def calculate_image_normalization_params(self):
        """(mins, maxes, means, stds), one entry per channel (general_dataset.py:206-220)."""
        s = channel_stats(self.entire_dataset["tactile_image"]).cpu()
        return (s[:, 0].tolist(), s[:, 1].tolist(), s[:, 2].tolist(), s[:, 3].tolist())

    # ---- samples ------------------------------------------------------------------------------------------------
    def _affines(self):
        if self._affine_cache is None:
            tA, tB = tactile_affine(self.image_normalization_method, self.norm_scale, self.image_normalization_parameters)
            dA, dB = depth_norm_affine(self.depth_normalization_method, self.norm_scale, self.depth_normalization_parameters)
            f = lambda v: torch.tensor(list(v), device=self.device, dtype=torch.float32)  # noqa: E731
            self._affine_cache = (f(tA), f(tB), f([dA]), f([dB]))
        return self._affine_cache

    def batch(self, idx: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Normalised samples for a device int64 index vector: the collated result of [self[i] for i in idx]."""
        idx = idx.to(self.device, dtype=torch.int64).contiguous()
        m = len(self)
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= m):
            raise IndexError(f"index out of range for a dataset of {m} samples")
        tA, tB, dA, dB = self._affines()
        return {"tactile_image": gather_affine(self.entire_dataset["tactile_image"], idx, tA, tB),
                "depth_image": gather_affine(self.entire_dataset["depth_image"], idx, dA, dB),
                "object_index": self.entire_dataset["object_index"][idx]}

    def __len__(self) -> int:
        return self.entire_dataset["tactile_image"].shape[0]

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        if idx < 0:
            idx += len(self)
        b = self.batch(torch.tensor([idx], dtype=torch.int64))
        return {k: v[0] for k, v in b.items()}


class DeviceLoader:
    """DataLoader(dataset, batch_size, shuffle, drop_last=False) for a DeviceDataset (train_unet.py:229-233).

    Batches are dicts of device tensors.  With shuffle=True the permutation is drawn the way torch's RandomSampler does
    under a DataLoader iterator (two draws from the global CPU generator -- the iterator's base seed, then the sampler's
    seed -- and torch.randperm under a private generator seeded with the latter), so under the same torch.manual_seed the
    sample order equals the reference loader's (checked against torch's own DataLoader in tests/test_oracle.py)."""

    def __init__(self, dataset: DeviceDataset, batch_size: int = 1, shuffle: bool = False, drop_last: bool = False,
                 rank: int = 0, world_size: int = 1) -> None:
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.rank, self.world_size = rank, world_size

    def unsharded(self) -> "DeviceLoader":
        """The same loader as ONE process sees it: the global batches (batch_size * world_size samples each, ragged tail as
        DataLoader(drop_last=False) leaves it), no wrap-around padding.  harness.fit evaluates validation / test passes
        through it on every rank: an evaluation issues no collective, and a padded shard would count the wrapped samples
        twice in the loss that early stopping reads."""
        if self.world_size == 1:
            return self
        return DeviceLoader(self.dataset, self.batch_size * self.world_size, self.shuffle, self.drop_last, rank=0, world_size=1)

    def eval_shares(self) -> Iterator[Tuple[Optional[Dict[str, torch.Tensor]], int, int]]:
        """Evaluation walk of the GLOBAL batches (batch_size * world_size samples, ragged tail as DataLoader(drop_last=False)
        leaves it): yields (batch, valid, global_count) per global batch, where `batch` holds this rank's contiguous share of
        it -- samples [rank*batch_size, (rank+1)*batch_size) of the global batch, NO wrap-around padding -- or None when the
        ragged tail leaves this rank nothing.  A short share is padded up to batch_size by repeating its last sample, so every
        forward keeps the per-rank train shape (no activation buffer is reallocated); only the first `valid` samples count.
        harness.evaluate_loader sums per-sample losses over them and all-reduces (sum, count) per global batch: every rank
        gets the single-process value without evaluating the whole set."""
        perm = self.order().to(self.dataset.device)
        n, bs, per = perm.numel(), self.batch_size, self.batch_size * self.world_size
        for s in range(0, n, per):
            g = perm[s:s + per]
            if g.numel() < per and self.drop_last:
                break
            mine = g[self.rank * bs:(self.rank + 1) * bs]
            valid = int(mine.numel())
            if valid == 0:
                yield None, 0, int(g.numel())
                continue
            if valid < bs:
                mine = torch.cat([mine, mine[-1:].expand(bs - valid)])
            yield self.dataset.batch(mine), valid, int(g.numel())

    def order(self) -> torch.Tensor:
        n = len(self.dataset)
        if not self.shuffle:
            return torch.arange(n, dtype=torch.int64)
        torch.empty((), dtype=torch.int64).random_()      # DataLoader's iterator draws its worker base seed first
        seed = int(torch.empty((), dtype=torch.int64).random_().item())     # RandomSampler.__iter__
        g = torch.Generator()
        g.manual_seed(seed)
        return torch.randperm(n, generator=g)

    def __len__(self) -> int:
        n = len(self.dataset)
        per = self.batch_size * self.world_size
        return n // per if self.drop_last else (n + per - 1) // per

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        """Data parallel: every rank draws the same permutation (same seed) and takes its contiguous share of each global
        batch of batch_size*world_size samples (contiguous split of the global batch).

        Every rank yields the SAME number of batches, each of the same size on every rank (the step issues collectives:
        a rank that skipped the last step would leave the others blocked in the gradient all-reduce, and unequal shares
        would mis-weight the 1/world gradient mean): a ragged last global batch is padded to a multiple of world_size by
        wrapping around to the start of the permutation, as torch's DistributedSampler does, or dropped with drop_last."""
        perm = self.order().to(self.dataset.device)
        n, per, world = perm.numel(), self.batch_size * self.world_size, self.world_size
        for s in range(0, n, per):
            g = perm[s:s + per]
            if g.numel() < per and self.drop_last:
                break
            pad = (-g.numel()) % world
            if pad:
                g = torch.cat([g, perm[torch.arange(pad, device=perm.device) % n]])
            share = g.numel() // world
            yield self.dataset.batch(g[self.rank * share:(self.rank + 1) * share])


def train_epoch(step, loader: DeviceLoader) -> Tuple[float, int]:
    """One pass of the reference's inner loop (train_unet.py:340-377) fed from HBM: returns (sum of batch losses, batches).
    The loss values stay on the device until the epoch ends (the reference syncs twice per step, :371 and :377)."""
    losses = []
    for data in loader:
        losses.append(step(data["tactile_image"], data["depth_image"]).detach().clone())   # the step reuses its loss buffer
    if not losses:
        return 0.0, 0
    stack = torch.stack(losses)
    if getattr(step, "nan_policy", None) is not None:
        # a skipped step's loss is NaN: the reference counts a NaN loss as 0.0 (train_unet.py:371-372), and so does
        # evaluate_loader -- one bad batch must not turn the epoch's (and, through the rank mean, every rank's) loss into NaN
        stack = torch.nan_to_num(stack, nan=0.0)
    total = float(stack.sum().item())
    if hasattr(step, "check_finite"):
        step.check_finite()          # nan_policy="raise": the epoch's one host sync has just happened
    return total, len(losses)
