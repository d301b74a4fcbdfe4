"""GPU parity of the two-dimensional Winograd F(2x4,3x3) dW kernel (gelslim_depth_amd/csrc/gsd_wgrad_w2d.hip; the dW half of
aten::convolution_backward for /root/reference/gelslim_depth/models/unet.py:11,14) against oracle/unet_numpy.py, called through
the C ABI entry gsd_conv3x3_wgrad the way the engine calls it: deferred BatchNorm+ReLU activation segments with slack, the
two-segment decoder form with its F.pad offset, a row-pitched dy.

Small shapes that reach what the 13 network shapes of test_gpu_layer_shapes.py do not: odd heights (a tile row half outside the
image), widths that leave the last k-step with tiles past the image, every k-step shape (1x4, 2x2, 4x1 tiles), both block forms
(128 co x 32 ci and 64 co x 64 ci), a second segment narrower and shorter than the grid, more splits than k-steps.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import rel_l1

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gsd():
    from gelslim_depth_amd import _lib
    return _lib


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def slack_dev(gsd, a):
    t = gsd.slack_empty(a.shape, "cuda")
    t.copy_(torch.from_numpy(np.ascontiguousarray(a)))
    return t


def pitched(t):
    n, c, h, w = t.shape
    base = torch.zeros((n, c, h, (w + 3) // 4 * 4), device=t.device, dtype=t.dtype)
    base[..., :w] = t
    return base[..., :w]


def bcast(v):
    return v[None, :, None, None]


def build_case(gsd, rng, n, h, w, c0, c1, co, bn, up_hw=None):
    """-> (segments, keep-alive list, activation as the conv sees it (numpy), dy (numpy), pitched dy tensor)"""
    from oracle import unet_numpy as on
    raw0 = rng.standard_normal((n, c0, h, w), dtype=np.float32)
    keep = []
    if bn:
        sc, sh = rng.uniform(0.5, 1.5, c0).astype(np.float32), (rng.standard_normal(c0) * 0.3).astype(np.float32)
        a0 = np.maximum(raw0 * bcast(sc) + bcast(sh), 0)
        r0d, scd, shd = slack_dev(gsd, raw0), dev(sc), dev(sh)
        segs = [gsd.make_src(r0d, scd, shd, relu=True, slack=gsd.SLACK)]
        keep += [r0d, scd, shd]
    else:
        a0 = raw0
        r0d = slack_dev(gsd, raw0)
        segs = [gsd.make_src(r0d, slack=gsd.SLACK)]
        keep.append(r0d)
    a = a0
    if c1:
        uh, uw = up_hw
        up = rng.standard_normal((n, c1, uh, uw), dtype=np.float32)
        upp, (top, left) = on.pad_to(up, h, w)
        a = np.concatenate([a0, upp], 1)
        upd = slack_dev(gsd, up)
        segs.append(gsd.make_src(upd, off=(top, left), slack=gsd.SLACK))
        keep.append(upd)
    dy = rng.standard_normal((n, co, h, w), dtype=np.float32)
    return segs, keep, a, dy, pitched(dev(dy))


def run_wgrad(gsd, segs, dyp, ci, co, n, h, w):
    L = gsd.lib
    need = L.gsd_conv3x3_wgrad_workspace(n, h, w, ci, co)
    ws = torch.zeros(need, device="cuda")
    dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
    src = gsd.src_array(segs)
    dy_src = gsd.make_src(dyp)
    form = L.gsd_conv3x3_wgrad_form(src, len(segs), C.byref(dy_src), ci, co, n, h, w)
    gsd.check(L.gsd_conv3x3_wgrad(src, len(segs), C.byref(dy_src), ci, co, dw.data_ptr(), ws.data_ptr(), need, n, h, w, gsd.stream_ptr()))
    torch.cuda.synchronize()
    return dw, form


def oracle_dw(a, dy, co, ci):
    from oracle import unet_numpy as on
    dwr = np.zeros((co, ci, 3, 3), np.float64)
    wdummy = np.zeros((co, ci, 3, 3), np.float32)
    for i in range(0, a.shape[0], 4):
        dwr += on.conv3x3_bwd(a[i:i + 4], wdummy, dy[i:i + 4], need_dx=False)[1]
    return dwr


# (n, h, w, c0, c1, co, deferred BatchNorm on segment 0, (up_h, up_w))
CASES = [
    (2, 37, 53, 32, 0, 128, True, None),      # odd height, 14 tile columns (1x4 k-steps leave 2 tiles past the image)
    (3, 20, 26, 64, 0, 64, False, None),      # 64 x 64 block form, plain source
    (2, 40, 53, 32, 32, 128, True, (38, 52)),  # two segments, the second shorter and narrower, F.pad offset (1, 0)
    (2, 24, 31, 64, 64, 64, True, (22, 30)),   # 64 x 64 block form with two segments
    (1, 8, 12, 64, 0, 256, True, None),       # fewer k-steps than blocks want splits
    (2, 33, 44, 96, 0, 128, True, None),      # W % 4 == 0: dy pitch == W; three n-blocks
    (5, 6, 9, 128, 0, 64, False, None),       # tiny image: every k-step is an edge k-step
    (2, 26, 45, 64, 0, 128, False, None),     # 128 x 32 block form, plain source (the instantiation of the encoder's first convs)
]


@pytest.mark.parametrize("case", CASES, ids=[f"N{c[0]}-{c[1]}x{c[2]}-C{c[3]}+{c[4]}-M{c[5]}" for c in CASES])
@pytest.mark.parametrize("kx", [0, 1, 2, 4])
def test_wgrad_w2d_vs_oracle(gsd, monkeypatch, case, kx):
    n, h, w, c0, c1, co, bn, up_hw = case
    ci = c0 + c1
    rng = np.random.default_rng(h * 1000 + w + kx)
    segs, keep, a, dy, dyp = build_case(gsd, rng, n, h, w, c0, c1, co, bn, up_hw)
    if kx:
        monkeypatch.setenv("GSD_WG2D_KX", str(kx))
    dw, form = run_wgrad(gsd, segs, dyp, ci, co, n, h, w)
    assert form == 2, "the two-dimensional form serves this call"
    got = dw.cpu().numpy()
    assert np.isfinite(got).all()
    ref = oracle_dw(a, dy, co, ci)
    assert rel_l1(got, ref) < 5e-5
    if c1:
        assert rel_l1(got[:, c0:], ref[:, c0:]) < 5e-5
    # run-to-run bitwise (ordered slab reduction)
    dw2, _ = run_wgrad(gsd, segs, dyp, ci, co, n, h, w)
    assert torch.equal(dw, dw2)
    # and the row form on the same operands agrees to rounding
    monkeypatch.setenv("GSD_WGRAD_W2D", "0")
    dw1, form1 = run_wgrad(gsd, segs, dyp, ci, co, n, h, w)
    assert form1 == 1
    assert rel_l1(dw1.cpu().numpy(), ref) < 5e-5
    del keep


def test_wgrad_w2d_declines_what_it_cannot_serve(gsd):
    """Unpitched dy with W % 4 != 0, a segment without slack, channel counts off its block grid: the row form serves the call."""
    L = gsd.lib
    rng = np.random.default_rng(3)
    n, h, w = 2, 12, 18
    for c0, co, slack, pitch in ((32, 128, True, False), (32, 128, False, True), (48, 128, True, True), (32, 96, True, True)):
        raw = rng.standard_normal((n, c0, h, w), dtype=np.float32)
        dy = rng.standard_normal((n, co, h, w), dtype=np.float32)
        rd = slack_dev(gsd, raw) if slack else dev(raw)
        segs = [gsd.make_src(rd, slack=gsd.SLACK if slack else 0)]
        dyp = pitched(dev(dy)) if pitch else dev(dy)
        if not pitch and not L.gsd_conv3x3_wgrad_takes_pitched_dy(n, h, w, c0, co):
            continue
        dw, form = run_wgrad(gsd, segs, dyp, c0, co, n, h, w)
        assert form == 1
        assert rel_l1(dw.cpu().numpy(), oracle_dw(raw, dy, co, c0)) < 5e-5


def test_wgrad_w2d_signed_bias_vs_fp64_at_inc_c1_batch32(gsd):
    """the Winograd dW's error against an fp64 dW must be NOISE, not bias.  inc.c1's shape (64 -> 64 @ 320x427,
    batch 32, deferred BatchNorm+ReLU source): the mean SIGNED error over all 36,864 weights is within 4 standard errors of zero
    and below a twentieth of the mean absolute error; the relative L1 is within the op tolerance."""
    n, h, w, c, co = 32, 320, 427, 64, 64
    g = torch.Generator(device="cuda").manual_seed(11)
    raw = gsd.slack_empty((n, c, h, w), "cuda")
    raw.normal_(generator=g)
    sc = torch.empty(c, device="cuda").uniform_(0.5, 1.5, generator=g)
    sh = torch.empty(c, device="cuda").normal_(generator=g) * 0.3
    dy = torch.zeros((n, co, h, 428), device="cuda")
    dy[..., :w].normal_(generator=g)
    dyp = dy[..., :w]
    segs = [gsd.make_src(raw, sc, sh, relu=True, slack=gsd.SLACK)]
    dw, form = run_wgrad(gsd, segs, dyp, c, co, n, h, w)
    assert form == 2
    # fp64 reference on the GPU, image by image (torch here is the checker's BLAS, not the product path)
    ref = torch.zeros((co, c, 3, 3), device="cuda", dtype=torch.float64)
    for i in range(n):
        a = torch.relu(raw[i:i + 1] * sc[None, :, None, None] + sh[None, :, None, None]).double()
        ap = torch.nn.functional.pad(a, (1, 1, 1, 1))
        d = dyp[i].double()                                    # (co, h, w)
        for r in range(3):
            for s in range(3):
                ref[:, :, r, s] += torch.einsum("ohw,chw->oc", d, ap[0, :, r:r + h, s:s + w])
    err = (dw.double() - ref)
    scale = ref.abs().mean().item()
    mean_signed = err.mean().item()
    std_err = err.std().item() / np.sqrt(err.numel())
    assert (err.abs().sum() / ref.abs().sum()).item() < 5e-5
    assert abs(mean_signed) < 4 * std_err + 1e-9 * scale, (mean_signed, std_err)
    assert abs(mean_signed) < 5e-2 * err.abs().mean().item() + 1e-9 * scale, (mean_signed, err.abs().mean().item())
