"""GPU parity, op level: every libgsd entry point against the oracle (oracle/unet_numpy.py) on seeded
inputs with odd shapes.  All calls go through the C ABI (gelslim_depth_amd._lib).

Tolerances: fp32 MFMA accumulates products in k order, the oracle uses BLAS / fp64; the bound for a
K-term dot product is ~K*eps relative to sum|a*b|, far below the 1e-3 relative-L1 north-star tolerance.
TOL = 2e-5 relative L1 everywhere an op is compared on identical inputs."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import rel_l1

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope="module")
def gsd():
    from gelslim_depth_amd import _lib
    return _lib


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rnd(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def layout(gsd, mode, w, co, ci):
    wt = torch.zeros(gsd.lib.gsd_weight_layout_size(mode, co, ci), device="cuda")
    gsd.check(gsd.lib.gsd_weight_layout(mode, w.data_ptr(), co, ci, wt.data_ptr(), gsd.stream_ptr()))
    return wt


class ConvForm:
    """The three forms of the conv3x3 entry points: direct taps (gsd_conv3x3*), Winograd F(4,3) rows (gsd_conv3x3_w43*) and the
    two-dimensional Winograd F(2x4,3x3) (gsd_conv3x3_w2d*)."""
    def __init__(self, gsd, algo, cin=None, c0=None):
        L = gsd.lib
        self.conv = (L.gsd_conv3x3, L.gsd_conv3x3_w43, L.gsd_conv3x3_w2d)[algo]
        self.dgrad_bnrelu = (L.gsd_conv3x3_dgrad_bnrelu, L.gsd_conv3x3_w43_dgrad_bnrelu, L.gsd_conv3x3_w2d_dgrad_bnrelu)[algo]
        self.partial_rows = (L.gsd_conv3x3_partial_rows, L.gsd_conv3x3_w43_partial_rows, L.gsd_conv3x3_w2d_partial_rows)[algo]
        self.mode_f, self.mode_d = ((0, 1), (4, 5), (8, 9))[algo]
        # F(4,3) adds two roundings per operand in the transforms (constants up to 8 and 1/24); F(2,3) constants are 1 and 1/2
        self.tol = 1e-5 if algo else TOL
        if algo == 2 and cin is not None and not L.gsd_conv3x3_w2d_supported(cin, cin if c0 is None else c0):
            pytest.skip("the two-dimensional form wants channel counts that are multiples of 4 (gsd_conv3x3_w2d_supported)")


ALGOS = pytest.mark.parametrize("algo", [0, 1, 2], ids=["direct", "w43", "w2d"])


def test_mfma_lane_maps(gsd):
    rng = np.random.default_rng(0)
    a = rng.integers(-8, 9, (16, 4)).astype(np.float32)       # asymmetric integer operands: exact in fp32
    b = rng.integers(-8, 9, (4, 16)).astype(np.float32)
    out = torch.zeros(16, 16, device="cuda")
    ad, bd = dev(a), dev(b)      # keep the device tensors alive across the launch
    gsd.check(gsd.lib.gsd_selftest_mfma(ad.data_ptr(), bd.data_ptr(), out.data_ptr(), gsd.stream_ptr()))
    assert np.array_equal(out.cpu().numpy(), a @ b)


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 5, 7, 9, 11), (1, 3, 64, 20, 33), (2, 8, 130, 13, 17), (1, 64, 64, 40, 53),
                                         (3, 16, 200, 5, 4), (2, 12, 64, 33, 70), (2, 16, 16, 1, 1), (1, 16, 32, 2, 3),
                                         (1, 20, 16, 3, 70), (1, 16, 16, 1, 130)])
@ALGOS
def test_conv3x3_plain(gsd, algo, n, ci, co, h, w):
    from oracle import unet_numpy as on
    F = ConvForm(gsd, algo, ci)
    rng = np.random.default_rng(n * 1000 + ci)
    x, wt_ = rnd(rng, n, ci, h, w), rnd(rng, co, ci, 3, 3, scale=0.2)
    xd, wd = dev(x), dev(wt_)
    y = torch.full((n, co, h, w), float("nan"), device="cuda")
    rows = F.partial_rows(n, h, w, co)
    mpad = (co + 63) // 64 * 64
    part = torch.zeros(rows * 2 * mpad, device="cuda")
    src = gsd.src_array([gsd.make_src(xd)])
    dst = gsd.dst_array([gsd.make_dst(y)])
    gsd.check(F.conv(src, 1, layout(gsd, F.mode_f, wd, co, ci).data_ptr(), ci, co, dst, 1, part.data_ptr(), n, h, w,
                     gsd.stream_ptr()))
    ref = on.conv3x3_fwd(x, wt_)
    got = y.cpu().numpy()
    assert np.isfinite(got).all(), "every output element must be written"
    assert rel_l1(got, ref) < F.tol
    # BatchNorm partial sums -> (sum, sumsq) per channel
    sums = torch.zeros(65 * 2 * co, device="cuda", dtype=torch.float64)
    gsd.check(gsd.lib.gsd_bn_reduce_partials(part.data_ptr(), rows, mpad, co, sums.data_ptr(), gsd.stream_ptr()))
    s = sums[:2 * co].cpu().numpy()
    r64 = ref.astype(np.float64)
    np.testing.assert_allclose(s[:co], r64.sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[co:], (r64 * r64).sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("c0,c1,co", [(6, 5, 9), (8, 12, 20), (64, 32, 70)])
@ALGOS
def test_conv3x3_deferred_bn_two_segments_and_crop(gsd, algo, c0, c1, co):
    """Consumer-side fusion: relu(bn(.)) on load, channel concat of two segments, F.pad offsets (unet.py:46-48);
    producer-side: two destinations with the crop that is F.pad's backward.  (6, 5): a 4-channel chunk straddles the
    segments (general fills); (8, 12), (64, 32): every chunk lies in one segment -- the straight fills with the one
    segment switch per block and the re-written padding positions."""
    from oracle import unet_numpy as on
    F = ConvForm(gsd, algo, c0 + c1, c0)
    rng = np.random.default_rng(7)
    n, h, w = 2, 9, 11
    skip_raw = rnd(rng, n, c0, h, w)
    sc, sh = rng.uniform(0.5, 1.5, c0).astype(np.float32), rnd(rng, c0, scale=0.3)
    up = rnd(rng, n, c1, 6, 8)          # diffY=3, diffX=3 -> top=1,left=1
    wt_ = rnd(rng, co, c0 + c1, 3, 3, scale=0.2)
    a0 = np.maximum(skip_raw * sc[None, :, None, None] + sh[None, :, None, None], 0)
    upp, (top, left) = on.pad_to(up, h, w)
    assert (top, left) == (1, 1)
    ref = on.conv3x3_fwd(np.concatenate([a0, upp], 1), wt_)
    srd, upd = dev(skip_raw), dev(up)
    scd, shd = dev(sc), dev(sh)
    y = torch.zeros((n, co, h, w), device="cuda")
    src = gsd.src_array([gsd.make_src(srd, scd, shd, relu=True), gsd.make_src(upd, off=(top, left))])
    gsd.check(F.conv(src, 2, layout(gsd, F.mode_f, dev(wt_), co, c0 + c1).data_ptr(), c0 + c1, co,
                     gsd.dst_array([gsd.make_dst(y)]), 1, None, n, h, w, gsd.stream_ptr()))
    assert rel_l1(y.cpu().numpy(), ref) < F.tol
    # dgrad with split destinations: dX of the same conv, routed to (skip grad | cropped up grad)
    if algo == 2 and not gsd.lib.gsd_conv3x3_w2d_supported(co, co):
        return          # the dX launch contracts over co channels: 9 and 70 are no multiples of 4 (the (8, 12, 20) case covers it)
    dy = rnd(rng, n, co, h, w)
    dxr, _ = on.conv3x3_bwd(np.concatenate([a0, upp], 1), wt_, dy)
    g_skip = torch.zeros((n, c0, h, w), device="cuda")
    g_up = torch.full((n, c1, 6, 8), float("nan"), device="cuda")
    dyd = dev(dy)
    gsd.check(F.conv(gsd.src_array([gsd.make_src(dyd)]), 1, layout(gsd, F.mode_d, dev(wt_), co, c0 + c1).data_ptr(),
                     co, c0 + c1, gsd.dst_array([gsd.make_dst(g_skip), gsd.make_dst(g_up, off=(top, left))]), 2,
                     None, n, h, w, gsd.stream_ptr()))
    assert rel_l1(g_skip.cpu().numpy(), dxr[:, :c0]) < F.tol
    assert rel_l1(g_up.cpu().numpy(), dxr[:, c0:, top:top + 6, left:left + 8]) < F.tol


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 5, 7, 9, 11), (1, 64, 64, 21, 29), (2, 20, 130, 6, 70), (1, 3, 16, 40, 53),
                                         (2, 16, 16, 1, 1), (1, 16, 32, 2, 3), (1, 32, 16, 3, 70), (3, 16, 200, 1, 130)])
def test_conv3x3_wgrad(gsd, n, ci, co, h, w):
    from oracle import unet_numpy as on
    rng = np.random.default_rng(ci * 31 + co)
    raw = rnd(rng, n, ci, h, w)
    sc, sh = rng.uniform(0.5, 1.5, ci).astype(np.float32), rnd(rng, ci, scale=0.3)
    a = np.maximum(raw * sc[None, :, None, None] + sh[None, :, None, None], 0)
    dy = rnd(rng, n, co, h, w)
    wdummy = np.zeros((co, ci, 3, 3), np.float32)
    _, dwr = on.conv3x3_bwd(a, wdummy, dy, need_dx=False)
    rawd, scd, shd, dyd = dev(raw), dev(sc), dev(sh), dev(dy)
    dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
    need = gsd.lib.gsd_conv3x3_wgrad_workspace(n, h, w, ci, co)
    ws = torch.zeros(need, device="cuda")
    a_src = gsd.src_array([gsd.make_src(rawd, scd, shd, relu=True)])
    dy_src = gsd.make_src(dyd)
    gsd.check(gsd.lib.gsd_conv3x3_wgrad(a_src, 1, C.byref(dy_src), ci, co, dw.data_ptr(), ws.data_ptr(), need, n, h, w,
                                        gsd.stream_ptr()))
    assert rel_l1(dw.cpu().numpy(), dwr) < 5e-5
    # too-small workspace is refused, not overrun
    rc = gsd.lib.gsd_conv3x3_wgrad(a_src, 1, C.byref(dy_src), ci, co, dw.data_ptr(), ws.data_ptr(), need - 1, n, h, w,
                                   gsd.stream_ptr())
    assert rc == -4 and b"workspace" in gsd.lib.gsd_last_error()


@pytest.mark.parametrize("n,c0,c1,co,h,w,uh,uw", [(2, 6, 5, 9, 9, 11, 6, 8), (1, 16, 16, 32, 21, 29, 20, 28),
                                                  (2, 64, 64, 64, 13, 19, 12, 18), (1, 24, 40, 130, 10, 37, 8, 36)])
def test_conv3x3_wgrad_two_segments(gsd, n, c0, c1, co, h, w, uh, uw):
    """dW of the decoder's first conv: activation = cat(relu(bn(skip)), F.pad(up)) (unet.py:46-48) given as two segments --
    a deferred-BatchNorm one and a plain one of another size placed at the F.pad offset.  Shapes with >= 16 channels run
    the Winograd form, the first one the direct form."""
    from oracle import unet_numpy as on
    rng = np.random.default_rng(c0 * 7 + c1)
    skip_raw = rnd(rng, n, c0, h, w)
    sc, sh = rng.uniform(0.5, 1.5, c0).astype(np.float32), rnd(rng, c0, scale=0.3)
    up = rnd(rng, n, c1, uh, uw)
    a0 = np.maximum(skip_raw * sc[None, :, None, None] + sh[None, :, None, None], 0)
    upp, (top, left) = on.pad_to(up, h, w)
    dy = rnd(rng, n, co, h, w)
    _, dwr = on.conv3x3_bwd(np.concatenate([a0, upp], 1), np.zeros((co, c0 + c1, 3, 3), np.float32), dy, need_dx=False)
    srd, upd, scd, shd, dyd = dev(skip_raw), dev(up), dev(sc), dev(sh), dev(dy)
    dw = torch.full((co, c0 + c1, 3, 3), float("nan"), device="cuda")
    need = gsd.lib.gsd_conv3x3_wgrad_workspace(n, h, w, c0 + c1, co)
    ws = torch.zeros(need, device="cuda")
    a_src = gsd.src_array([gsd.make_src(srd, scd, shd, relu=True), gsd.make_src(upd, off=(top, left))])
    dy_src = gsd.make_src(dyd)
    gsd.check(gsd.lib.gsd_conv3x3_wgrad(a_src, 2, C.byref(dy_src), c0 + c1, co, dw.data_ptr(), ws.data_ptr(), need, n, h, w,
                                        gsd.stream_ptr()))
    got = dw.cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_l1(got, dwr) < 5e-5
    assert rel_l1(got[:, c0:], dwr[:, c0:]) < 5e-5      # the offset segment on its own


@pytest.mark.parametrize("n,ci,h,w", [(2, 8, 4, 5), (1, 128, 20, 26), (2, 36, 7, 9), (3, 24, 5, 53), (1, 264, 3, 2), (2, 16, 1, 1)])
def test_convT_fwd_bwd(gsd, monkeypatch, n, ci, h, w):
    from oracle import unet_numpy as on
    rng = np.random.default_rng(ci)
    co = ci // 2
    raw = rnd(rng, n, ci, h, w)
    sc, sh = rng.uniform(0.5, 1.5, ci).astype(np.float32), rnd(rng, ci, scale=0.3)
    x = np.maximum(raw * sc[None, :, None, None] + sh[None, :, None, None], 0)
    wt_, b = rnd(rng, ci, co, 2, 2, scale=0.2), rnd(rng, co)
    ref = on.convT_fwd(x, wt_, b)
    rawd, scd, shd, wd, bd = dev(raw), dev(sc), dev(sh), dev(wt_), dev(b)
    y = torch.full((n, co, 2 * h, 2 * w), float("nan"), device="cuda")
    s = gsd.make_src(rawd, scd, shd, relu=True)
    d = gsd.make_dst(y)
    gsd.check(gsd.lib.gsd_convT2x2(C.byref(s), layout(gsd, 6, wd, co, ci).data_ptr(), bd.data_ptr(), ci, co, C.byref(d), n, h,
                                   w, gsd.stream_ptr()))
    assert rel_l1(y.cpu().numpy(), ref) < TOL
    dy = rnd(rng, n, co, 2 * h, 2 * w)
    dxr, dwr, dbr = on.convT_bwd(x, wt_, dy)
    dyd = dev(dy)
    sdy = gsd.make_src(dyd)
    # dX: the LDS-DMA kernel (weight layout 7; an odd-width plane needs 2 readable floats behind dy: gsd_src.slack) and the
    # register-staged kernel (layout 3) -- gsd_convT2x2_dgrad_layout says which one a call will run
    dys = gsd.slack_empty(dyd.shape, "cuda")
    dys.copy_(dyd)
    for src_, env in ((gsd.make_src(dys, slack=gsd.SLACK), None), (sdy, "0")):
        if env is not None:
            monkeypatch.setenv("GSD_CONVT_DG_DMA", env)
        mode = gsd.lib.gsd_convT2x2_dgrad_layout(C.byref(src_), ci, co, n, h, w)
        assert mode == (7 if env is None else 3)
        dx = torch.full((n, ci, h, w), float("nan"), device="cuda")
        ddx = gsd.make_dst(dx)
        gsd.check(gsd.lib.gsd_convT2x2_dgrad(C.byref(src_), layout(gsd, mode, wd, co, ci).data_ptr(), ci, co, C.byref(ddx), n, h, w,
                                             gsd.stream_ptr()))
        assert bool(torch.isfinite(dx).all())
        assert rel_l1(dx.cpu().numpy(), dxr) < TOL, mode
    monkeypatch.delenv("GSD_CONVT_DG_DMA")
    if w % 2:      # an odd width without slack falls back to the register-staged kernel instead of reading past the tensor
        assert gsd.lib.gsd_convT2x2_dgrad_layout(C.byref(sdy), ci, co, n, h, w) == 3
        # ... and a caller that STATES a mode-7 image for such arguments is refused, not handed the other kernel's layout
        dx = torch.zeros((n, ci, h, w), device="cuda")
        rc = gsd.lib.gsd_convT2x2_dgrad_as(7, C.byref(sdy), layout(gsd, 7, wd, co, ci).data_ptr(), ci, co, C.byref(gsd.make_dst(dx)),
                                           n, h, w, gsd.stream_ptr())
        assert rc != 0 and "mode-7" in gsd.lib.gsd_last_error().decode()
    # the stated-mode entry runs either kernel on demand, whatever the environment says (the engine lays its image out once)
    monkeypatch.setenv("GSD_CONVT_DG_DMA", "0")
    for mode in (7, 3):
        dx = torch.full((n, ci, h, w), float("nan"), device="cuda")
        gsd.check(gsd.lib.gsd_convT2x2_dgrad_as(mode, C.byref(gsd.make_src(dys, slack=gsd.SLACK)), layout(gsd, mode, wd, co, ci).data_ptr(),
                                                ci, co, C.byref(gsd.make_dst(dx)), n, h, w, gsd.stream_ptr()))
        assert rel_l1(dx.cpu().numpy(), dxr) < TOL, mode
    monkeypatch.delenv("GSD_CONVT_DG_DMA")
    need = gsd.lib.gsd_convT2x2_wgrad_workspace(n, h, w, ci, co)
    ws = torch.zeros(need, device="cuda")
    dw = torch.full((ci, co, 2, 2), float("nan"), device="cuda")
    db = torch.full((co,), float("nan"), device="cuda")
    gsd.check(gsd.lib.gsd_convT2x2_wgrad(C.byref(s), C.byref(sdy), ci, co, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), need,
                                         n, h, w, gsd.stream_ptr()))
    assert rel_l1(dw.cpu().numpy(), dwr) < 5e-5
    assert rel_l1(db.cpu().numpy(), dbr) < TOL


@pytest.mark.parametrize("n,ci,h,w", [(2, 8, 4, 5), (1, 128, 20, 26), (2, 36, 7, 9), (3, 24, 5, 53), (1, 264, 3, 2), (2, 16, 1, 1),
                                       (2, 192, 12, 11)])
def test_convT_dgrad_bnrelu_is_the_unfused_pair(gsd, n, ci, h, w):
    """gsd_convT2x2_dgrad_bnrelu: dz bit-equal to the plain dX launch masked by relu'(bn(raw)); the partial rows sum to
    (sum dz, sum dz * xhat) per channel (fp64 check), pad channels of a row are zero."""
    rng = np.random.default_rng(ci + w)
    co = ci // 2
    raw = rnd(rng, n, ci, h, w)
    sc, sh = rng.uniform(0.5, 1.5, ci).astype(np.float32), rnd(rng, ci, scale=0.3)
    mu, inv = rnd(rng, ci, scale=0.2), rng.uniform(0.5, 2.0, ci).astype(np.float32)
    wt_ = rnd(rng, ci, co, 2, 2, scale=0.2)
    dy = rnd(rng, n, co, 2 * h, 2 * w)
    dys = gsd.slack_empty(dy.shape, "cuda")
    dys.copy_(dev(dy))
    src_ = gsd.make_src(dys, slack=gsd.SLACK)
    wimg = layout(gsd, 7, dev(wt_), co, ci)
    rows = gsd.lib.gsd_convT2x2_dgrad_bnrelu_partial_rows(C.byref(src_), ci, co, n, h, w)
    assert rows == 2 * -(-(n * h * (w + (w & 1))) // 128)
    dx = torch.full((n, ci, h, w), float("nan"), device="cuda")
    gsd.check(gsd.lib.gsd_convT2x2_dgrad_as(7, C.byref(src_), wimg.data_ptr(), ci, co, C.byref(gsd.make_dst(dx)), n, h, w, gsd.stream_ptr()))
    rawd, scd, shd, mud, invd = dev(raw), dev(sc), dev(sh), dev(mu), dev(inv)
    mp = -(-ci // 64) * 64
    part = torch.full((rows, 2 * mp), float("nan"), device="cuda")
    dz = torch.full((n, ci, h, w), float("nan"), device="cuda")
    gsd.check(gsd.lib.gsd_convT2x2_dgrad_bnrelu(C.byref(src_), wimg.data_ptr(), ci, co, C.byref(gsd.make_dst(dz)), rawd.data_ptr(),
                                                scd.data_ptr(), shd.data_ptr(), mud.data_ptr(), invd.data_ptr(), part.data_ptr(),
                                                n, h, w, gsd.stream_ptr()))
    mask = torch.addcmul(shd[None, :, None, None], rawd, scd[None, :, None, None]) > 0    # one fma, as the kernel's
    want = torch.where(mask, dx, torch.zeros_like(dx))
    assert torch.equal(dz, want)
    p = part.cpu().numpy().astype(np.float64)
    assert np.isfinite(p).all()
    assert not p[:, ci:mp].any() and not p[:, mp + ci:].any()
    z = want.cpu().numpy().astype(np.float64)
    xhat = (raw.astype(np.float64) - mu[None, :, None, None]) * inv[None, :, None, None]
    s1, s2 = z.sum(axis=(0, 2, 3)), (z * xhat).sum(axis=(0, 2, 3))
    assert np.abs(p[:, :ci].sum(0) - s1).max() <= 2e-5 * max(1.0, np.abs(z).sum(axis=(0, 2, 3)).max())
    assert np.abs(p[:, mp:mp + ci].sum(0) - s2).max() <= 2e-5 * max(1.0, np.abs(z * xhat).sum(axis=(0, 2, 3)).max())
    # an odd-width gradient without the readable floats behind it is refused (the kernel reads pixel pairs)
    if w % 2:
        plain = gsd.make_src(dev(dy))
        assert gsd.lib.gsd_convT2x2_dgrad_bnrelu_partial_rows(C.byref(plain), ci, co, n, h, w) == 0
        rc = gsd.lib.gsd_convT2x2_dgrad_bnrelu(C.byref(plain), wimg.data_ptr(), ci, co, C.byref(gsd.make_dst(dz)), rawd.data_ptr(),
                                               scd.data_ptr(), shd.data_ptr(), mud.data_ptr(), invd.data_ptr(), part.data_ptr(),
                                               n, h, w, gsd.stream_ptr())
        assert rc != 0


def test_bn_finalize_and_eval_coeffs(gsd):
    from oracle import unet_numpy as on
    rng = np.random.default_rng(3)
    n, c, h, w = 3, 5, 7, 9
    x = rnd(rng, n, c, h, w) * 2 + 1
    g, b = rng.uniform(0.5, 1.5, c).astype(np.float32), rnd(rng, c)
    rm, rv = rnd(rng, c, scale=0.1), rng.uniform(0.5, 1.5, c).astype(np.float32)
    yr, (mr, ir), (nrm, nrv) = on.bn_train_fwd(x, g, b, rm, rv)
    x64 = x.astype(np.float64)
    sums = torch.zeros(65 * 2 * c, dtype=torch.float64, device="cuda")
    sums[:c] = torch.from_numpy(x64.sum(axis=(0, 2, 3))).cuda()
    sums[c:2 * c] = torch.from_numpy((x64 * x64).sum(axis=(0, 2, 3))).cuda()
    gd, bd, rmd, rvd = dev(g), dev(b), dev(rm), dev(rv)
    outs = [torch.zeros(c, device="cuda") for _ in range(4)]
    gsd.check(gsd.lib.gsd_bn_finalize(sums.data_ptr(), c, float(n * h * w), gd.data_ptr(), bd.data_ptr(), 1e-5, 0.1,
                                      rmd.data_ptr(), rvd.data_ptr(), *[o.data_ptr() for o in outs], None, gsd.stream_ptr()))
    mean, invstd, scale, shift = [o.cpu().numpy() for o in outs]
    assert rel_l1(mean, mr) < 1e-6 and rel_l1(invstd, ir) < 1e-6
    assert rel_l1(rmd.cpu().numpy(), nrm) < 1e-6 and rel_l1(rvd.cpu().numpy(), nrv) < 1e-6
    y = x * scale[None, :, None, None] + shift[None, :, None, None]
    assert rel_l1(y, yr) < 1e-6
    sc2, sh2 = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    gsd.check(gsd.lib.gsd_bn_eval_coeffs(gd.data_ptr(), bd.data_ptr(), rmd.data_ptr(), rvd.data_ptr(), 1e-5, c,
                                         sc2.data_ptr(), sh2.data_ptr(), gsd.stream_ptr()))
    ye = x * sc2.cpu().numpy()[None, :, None, None] + sh2.cpu().numpy()[None, :, None, None]
    assert rel_l1(ye, on.bn_eval_fwd(x, g, b, nrm, nrv)) < 1e-6


def _bn_setup(rng, n, c, h, w):
    from oracle import unet_numpy as on
    raw = rnd(rng, n, c, h, w) + 0.3
    g, b = rng.uniform(0.5, 1.5, c).astype(np.float32), rnd(rng, c, scale=0.3)
    y, (mean, invstd), _ = on.bn_train_fwd(raw, g, b, np.zeros(c, np.float32), np.ones(c, np.float32))
    scale = (g * invstd).astype(np.float32)
    shift = (b - mean * scale).astype(np.float32)
    return raw, g, b, mean, invstd, scale, shift, np.maximum(y, 0)


def _bn_bwd_run(gsd, mode, raw, scale, shift, mean, invstd, n, c, h, w, da=None, dpool=None, dout=None, wout=None):
    rawd = dev(raw)
    vecs = [dev(v) for v in (scale, shift, mean, invstd)]
    g = dev(da) if da is not None else torch.zeros((n, c, h, w), device="cuda")
    rows = gsd.lib.gsd_bn_bwd_partial_rows(n, c, h, w)
    part = torch.zeros(rows * 3 * c, device="cuda")
    das = gsd.make_src(g)
    dp = dev(dpool) if dpool is not None else None
    do = dev(dout) if dout is not None else None
    wo = dev(wout) if wout is not None else None
    gsd.check(gsd.lib.gsd_bn_bwd_reduce(mode, rawd.data_ptr(), *[v.data_ptr() for v in vecs], C.byref(das), gsd.ptr(dp),
                                        gsd.ptr(do), gsd.ptr(wo), 1, g.data_ptr(), part.data_ptr(), n, c, h, w,
                                        gsd.stream_ptr()))
    sums = torch.zeros(65 * 3 * c, dtype=torch.float64, device="cuda")
    gsd.check(gsd.lib.gsd_bn_bwd_reduce_partials(part.data_ptr(), rows, c, sums.data_ptr(), gsd.stream_ptr()))
    outs = [torch.zeros(c, device="cuda") for _ in range(5)]   # dgamma dbeta dwout c1 c2
    gsd.check(gsd.lib.gsd_bn_bwd_finalize(sums.data_ptr(), None, c, float(n * h * w), *[o.data_ptr() for o in outs],
                                          gsd.stream_ptr()))
    # out-of-place form into a row-pitched buffer (what the engine feeds the Winograd dW / dX kernels), then the in-place form:
    # identical values, zeros in the pitch padding, the input left untouched by the first call
    pitch = (w + 3) // 4 * 4
    base = torch.full((n, c, h, pitch), float("nan"), device="cuda")
    g_before = g.clone()
    args = (rawd.data_ptr(), vecs[0].data_ptr(), vecs[2].data_ptr(), vecs[3].data_ptr(), outs[3].data_ptr(), outs[4].data_ptr(),
            n, c, h, w)
    gsd.check(gsd.lib.gsd_bn_bwd_apply(g.data_ptr(), *args, base.data_ptr(), pitch, gsd.stream_ptr()))
    assert torch.equal(g, g_before)
    gsd.check(gsd.lib.gsd_bn_bwd_apply(g.data_ptr(), *args, None, 0, gsd.stream_ptr()))
    assert torch.equal(base[..., :w], g) and bool((base[..., w:] == 0).all())
    return g.cpu().numpy(), [o.cpu().numpy() for o in outs]


def test_bn_relu_backward_plain(gsd):
    from oracle import unet_numpy as on
    rng = np.random.default_rng(5)
    n, c, h, w = 3, 5, 7, 9
    raw, g, b, mean, invstd, scale, shift, a = _bn_setup(rng, n, c, h, w)
    da = rnd(rng, n, c, h, w)
    dxr, dgr, dbr = on.bn_train_bwd(raw, g, mean, invstd, da * (a > 0))
    dx, (dg, db, _, _, _) = _bn_bwd_run(gsd, 0, raw, scale, shift, mean, invstd, n, c, h, w, da=da)
    assert rel_l1(dx, dxr) < TOL and rel_l1(dg, dgr) < TOL and rel_l1(db, dbr) < TOL


def test_bn_relu_maxpool_backward(gsd):
    """mode POOL: skip gradient + max-pool routed gradient, arg-max recomputed (first max wins), odd H/W."""
    from oracle import unet_numpy as on
    rng = np.random.default_rng(6)
    n, c, h, w = 2, 4, 9, 11
    raw, g, b, mean, invstd, scale, shift, a = _bn_setup(rng, n, c, h, w)
    a_gpu_like = np.maximum(raw * scale[None, :, None, None] + shift[None, :, None, None], 0)
    _, idx = on.maxpool2_fwd(a_gpu_like)
    dpool = rnd(rng, n, c, h // 2, w // 2)
    dskip = rnd(rng, n, c, h, w)
    da = dskip + on.maxpool2_bwd(dpool, idx, raw.shape)
    dxr, dgr, dbr = on.bn_train_bwd(raw, g, mean, invstd, da * (a > 0))
    dx, (dg, db, _, _, _) = _bn_bwd_run(gsd, 1, raw, scale, shift, mean, invstd, n, c, h, w, da=dskip, dpool=dpool)
    assert rel_l1(dx, dxr) < TOL and rel_l1(dg, dgr) < TOL and rel_l1(db, dbr) < TOL


def test_outconv_forward_loss_and_backward(gsd):
    from oracle import unet_numpy as on
    rng = np.random.default_rng(8)
    n, c, h, w = 2, 6, 7, 9
    raw, g, b, mean, invstd, scale, shift, a = _bn_setup(rng, n, c, h, w)
    wout, bout = rnd(rng, 1, c, 1, 1, scale=0.4), rnd(rng, 1)
    tgt = rnd(rng, n, 1, h, w)
    yr = on.conv1x1_fwd(a, wout, bout)
    rawd, scd, shd = dev(raw), dev(scale), dev(shift)
    y = torch.zeros((n, 1, h, w), device="cuda")
    s = gsd.make_src(rawd, scd, shd, relu=True)
    woutd, boutd = dev(wout), dev(bout)
    gsd.check(gsd.lib.gsd_conv1x1_out(C.byref(s), woutd.data_ptr(), boutd.data_ptr(), c, 1, y.data_ptr(), n, h, w,
                                      gsd.stream_ptr()))
    assert rel_l1(y.cpu().numpy(), yr) < TOL
    for kind, fn in ((0, on.mse_loss), (1, on.l1_loss)):
        lr, gr = fn(yr, tgt)
        loss = torch.zeros(1, device="cuda")
        grad = torch.zeros((n, 1, h, w), device="cuda")
        ws = torch.zeros(2048, dtype=torch.float64, device="cuda")
        yrd, tgtd = dev(yr), dev(tgt)
        gsd.check(gsd.lib.gsd_loss_fwd_bwd(kind, yrd.data_ptr(), tgtd.data_ptr(), yr.size, 1.0, loss.data_ptr(),
                                           grad.data_ptr(), ws.data_ptr(), None, gsd.stream_ptr()))
        assert abs(loss.item() - lr) < 1e-6 * abs(lr)
        assert rel_l1(grad.cpu().numpy(), gr) < 1e-6
    _, dout = on.mse_loss(yr, tgt)
    dar, dwr, dbr = on.conv1x1_bwd(a, wout, dout)
    dxr, dgr, dbtr = on.bn_train_bwd(raw, g, mean, invstd, dar * (a > 0))
    dx, (dg, db, dwo, _, _) = _bn_bwd_run(gsd, 2, raw, scale, shift, mean, invstd, n, c, h, w, dout=dout,
                                          wout=wout.reshape(1, c))
    assert rel_l1(dx, dxr) < TOL and rel_l1(dg, dgr) < TOL and rel_l1(db, dbtr) < TOL
    assert rel_l1(dwo, dwr.reshape(-1)) < TOL
    out = torch.zeros(1, device="cuda")
    ws = torch.zeros(64, device="cuda")
    doutd = dev(dout)
    gsd.check(gsd.lib.gsd_sum_planes(doutd.data_ptr(), n, 1, h * w, out.data_ptr(), ws.data_ptr(), gsd.stream_ptr()))
    assert rel_l1(out.cpu().numpy(), dbr) < TOL


def test_maxpool_floor(gsd):
    from oracle import unet_numpy as on
    rng = np.random.default_rng(9)
    n, c, h, w = 2, 4, 9, 11
    raw = rnd(rng, n, c, h, w)
    sc, sh = rng.uniform(0.5, 1.5, c).astype(np.float32), rnd(rng, c, scale=0.3)
    a = np.maximum(raw * sc[None, :, None, None] + sh[None, :, None, None], 0)
    yr, _ = on.maxpool2_fwd(a)
    rawd, scd, shd = dev(raw), dev(sc), dev(sh)
    y = torch.zeros((n, c, h // 2, w // 2), device="cuda")
    s = gsd.make_src(rawd, scd, shd, relu=True)
    gsd.check(gsd.lib.gsd_maxpool2(C.byref(s), y.data_ptr(), n, c, h, w, gsd.stream_ptr()))
    assert np.allclose(y.cpu().numpy(), yr, rtol=1e-6, atol=1e-6)


def test_adam_ema_matches_torch_and_oracle(gsd):
    from oracle import unet_numpy as on
    rng = np.random.default_rng(10)
    numel = 10007
    p0 = rnd(rng, numel)
    p, m, v = p0.copy(), np.zeros(numel, np.float32), np.zeros(numel, np.float32)
    shadow = p0.copy()
    pd, md, vd, ed = dev(p0), torch.zeros(numel, device="cuda"), torch.zeros(numel, device="cuda"), dev(p0)
    pt = torch.from_numpy(p0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-3, weight_decay=1e-6)
    for step in range(1, 4):
        g = rnd(rng, numel, scale=0.1)
        on.adam_step(p, g, m, v, step)
        on.ema_update(shadow, p, step)
        pt.grad = torch.from_numpy(g.copy())
        opt.step()
        d = on.ema_decay(step)
        gd = dev(g)
        gsd.check(gsd.lib.gsd_adam_ema(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), ed.data_ptr(), numel,
                                       step, 1e-3, 0.9, 0.999, 1e-8, 1e-6, d, 1.0, None, gsd.stream_ptr()))
    assert rel_l1(pd.cpu().numpy(), pt.detach().numpy()) < 1e-6
    assert rel_l1(pd.cpu().numpy(), p) < 1e-6
    assert rel_l1(ed.cpu().numpy(), shadow) < 1e-6


def test_bad_arguments_are_refused(gsd):
    x = torch.zeros((1, 4, 8, 8), device="cuda")
    y = torch.zeros((1, 4, 8, 8), device="cuda")
    src = gsd.src_array([gsd.make_src(x)])
    dst = gsd.dst_array([gsd.make_dst(y)])
    wt = torch.zeros(gsd.lib.gsd_weight_layout_size(0, 4, 4), device="cuda")
    assert gsd.lib.gsd_conv3x3(src, 1, wt.data_ptr(), 5, 4, dst, 1, None, 1, 8, 8, gsd.stream_ptr()) == -1   # Cin mismatch
    assert gsd.lib.gsd_conv3x3(src, 1, None, 4, 4, dst, 1, None, 1, 8, 8, gsd.stream_ptr()) == -1
    assert gsd.lib.gsd_conv3x3(src, 3, wt.data_ptr(), 4, 4, dst, 1, None, 1, 8, 8, gsd.stream_ptr()) == -1
    with pytest.raises(gsd.GsdError):
        gsd.check(gsd.lib.gsd_weight_layout(10, x.data_ptr(), 4, 4, wt.data_ptr(), gsd.stream_ptr()), "weight_layout")


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 6, 9, 9, 11), (1, 64, 130, 21, 29), (2, 10, 12, 10, 13), (1, 70, 128, 21, 29)])
@ALGOS
def test_conv3x3_dgrad_fused_with_bn_relu_backward(gsd, algo, n, ci, co, h, w):
    """gsd_conv3x3_dgrad_bnrelu == conv dX followed by gsd_bn_bwd_reduce(mode PLAIN): dz and (sum dz, sum dz*xhat)."""
    from oracle import unet_numpy as on
    F = ConvForm(gsd, algo, co, co)       # the dX launch contracts over the unit's OUTPUT channels
    rng = np.random.default_rng(co)
    # forward unit "prev": raw (n, ci, h, w) with its BN; this conv maps ci -> co
    raw, g, b, mean, invstd, scale, shift, a = _bn_setup(rng, n, ci, h, w)
    wt_ = rnd(rng, co, ci, 3, 3, scale=0.2)
    dy = rnd(rng, n, co, h, w)
    da, _ = on.conv3x3_bwd(a, wt_, dy)
    dz_ref = da * (a > 0)
    xhat = (raw.astype(np.float64) - mean[None, :, None, None]) * invstd[None, :, None, None]
    s1_ref, s2_ref = dz_ref.astype(np.float64).sum(axis=(0, 2, 3)), (dz_ref * xhat).sum(axis=(0, 2, 3))
    dyd, rawd = dev(dy), dev(raw)
    vecs = [dev(v) for v in (scale, shift, mean, invstd)]
    dz = torch.full((n, ci, h, w), float("nan"), device="cuda")
    rows = F.partial_rows(n, h, w, ci)
    mpad = (ci + 63) // 64 * 64
    part = torch.zeros(rows * 2 * mpad, device="cuda")
    wl = layout(gsd, F.mode_d, dev(wt_), co, ci)
    s, d = gsd.make_src(dyd), gsd.make_dst(dz)
    gsd.check(F.dgrad_bnrelu(C.byref(s), wl.data_ptr(), co, ci, C.byref(d), rawd.data_ptr(),
                             *[v.data_ptr() for v in vecs], part.data_ptr(), n, h, w, gsd.stream_ptr()))
    assert rel_l1(dz.cpu().numpy(), dz_ref) < F.tol
    sums = torch.zeros(65 * 2 * ci, device="cuda", dtype=torch.float64)
    gsd.check(gsd.lib.gsd_bn_reduce_partials(part.data_ptr(), rows, mpad, ci, sums.data_ptr(), gsd.stream_ptr()))
    got = sums[:2 * ci].cpu().numpy()
    np.testing.assert_allclose(got[:ci], s1_ref, rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(got[ci:], s2_ref, rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize("rows,c,mpad", [(1, 3, 16), (37, 64, 64), (1024, 200, 256), (515, 1024, 1024)])
def test_bn_one_launch_reduce_finalize(gsd, rows, c, mpad):
    """gsd_bn_reduce_finalize / gsd_bn_bwd_reduce_finalize == the two-launch forms (fp64 sums in another order:
    identical after rounding to fp32 up to 1 ulp) and a float64 torch column sum."""
    rng = np.random.default_rng(rows + c)
    count = 1234.0
    # forward layout: rows of [sum | sum of squares] with ld = 2*mpad
    y = rng.standard_normal((rows, mpad)).astype(np.float32)
    part = dev(np.concatenate([y, y * y + 1.0], axis=1))
    gamma, beta = dev(rnd(rng, c)), dev(rnd(rng, c))
    rm0, rv0 = rnd(rng, c), np.abs(rnd(rng, c)) + 0.5

    def fwd(one):
        sums = torch.zeros(65 * 3 * c, dtype=torch.float64, device="cuda")
        rm, rv = dev(rm0), dev(rv0)
        outs = [torch.zeros(c, device="cuda") for _ in range(4)]
        tail = [gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr()] + [o.data_ptr() for o in outs]
        if one:
            gsd.check(gsd.lib.gsd_bn_reduce_finalize(part.data_ptr(), rows, mpad, c, sums.data_ptr(), count, *tail,
                                                     None, gsd.stream_ptr()))
        else:
            gsd.check(gsd.lib.gsd_bn_reduce_partials(part.data_ptr(), rows, mpad, c, sums.data_ptr(), gsd.stream_ptr()))
            gsd.check(gsd.lib.gsd_bn_finalize(sums.data_ptr(), c, count, *tail, None, gsd.stream_ptr()))
        torch.cuda.synchronize()
        return sums[:2 * c].cpu().numpy(), [o.cpu().numpy() for o in outs + [rm, rv]]

    s1, o1 = fwd(True)
    s2, o2 = fwd(False)
    ref = part.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(s1, np.concatenate([ref[:c], ref[mpad:mpad + c]]), rtol=1e-13, atol=1e-10)
    np.testing.assert_allclose(s1, s2, rtol=1e-13, atol=1e-10)
    for a, b in zip(o1, o2):
        np.testing.assert_allclose(a, b, rtol=3e-7, atol=1e-7)

    # backward, both layouts
    for layout_mpad in (0, mpad):
        ld = 2 * mpad if layout_mpad else 3 * c
        bp = dev(rnd(rng, rows, ld))

        def bwd(one):
            sums = torch.zeros(65 * 3 * c, dtype=torch.float64, device="cuda")
            outs = [torch.zeros(c, device="cuda") for _ in range(5)]   # dgamma dbeta dwout c1 c2
            ptrs = [o.data_ptr() for o in outs]
            if layout_mpad:
                ptrs[2] = None
            if one:
                gsd.check(gsd.lib.gsd_bn_bwd_reduce_finalize(bp.data_ptr(), rows, layout_mpad, c, sums.data_ptr(), count, *ptrs,
                                                             gsd.stream_ptr()))
            else:
                if layout_mpad:
                    gsd.check(gsd.lib.gsd_bn_reduce_partials(bp.data_ptr(), rows, mpad, c, sums.data_ptr(), gsd.stream_ptr()))
                else:
                    gsd.check(gsd.lib.gsd_bn_bwd_reduce_partials(bp.data_ptr(), rows, c, sums.data_ptr(), gsd.stream_ptr()))
                gsd.check(gsd.lib.gsd_bn_bwd_finalize(sums.data_ptr(), None, c, count, *ptrs, gsd.stream_ptr()))
            torch.cuda.synchronize()
            return [o.cpu().numpy() for o in outs]

        b1, b2 = bwd(True), bwd(False)
        col = bp.double().sum(0).cpu().numpy()
        off2 = mpad if layout_mpad else c
        np.testing.assert_allclose(b1[1], col[:c], rtol=3e-7, atol=1e-6)            # dbeta
        np.testing.assert_allclose(b1[0], col[off2:off2 + c], rtol=3e-7, atol=1e-6)  # dgamma
        for a, b in zip(b1, b2):
            np.testing.assert_allclose(a, b, rtol=3e-7, atol=1e-7)


def pitched(t, fill=0.0):
    """Copy of an (N,C,H,W) device tensor as a view of a row-pitched buffer (pitch = W rounded up to 4 floats); the pitch
    padding holds `fill` -- the operand's padding value, as the ABI asks (include/gsd.h: gsd_src.w_stride)."""
    n, c, h, w = t.shape
    base = torch.full((n, c, h, (w + 3) // 4 * 4), fill, device=t.device, dtype=t.dtype)
    base[..., :w] = t
    return base[..., :w]


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 16, 16, 9, 11), (1, 64, 64, 21, 29), (2, 20, 130, 6, 70), (1, 128, 64, 12, 427),
                                         (3, 32, 48, 5, 16), (2, 64, 128, 17, 53)])
def test_conv3x3_wgrad_pitched_dy(gsd, n, ci, co, h, w):
    """dy from a row-pitched buffer (what gsd_bn_bwd_apply's out-of-place form produces): the Winograd dW kernel moves it as
    aligned 16-byte pieces into an XOR-swizzled LDS image -- same result as from the contiguous tensor, bit for bit
    (same products, same summation order), and against the oracle."""
    from oracle import unet_numpy as on
    rng = np.random.default_rng(ci * 13 + co + w)
    raw = rnd(rng, n, ci, h, w)
    sc, sh = rng.uniform(0.5, 1.5, ci).astype(np.float32), rnd(rng, ci, scale=0.3)
    a = np.maximum(raw * sc[None, :, None, None] + sh[None, :, None, None], 0)
    dy = rnd(rng, n, co, h, w)
    _, dwr = on.conv3x3_bwd(a, np.zeros((co, ci, 3, 3), np.float32), dy, need_dx=False)
    rawd, scd, shd, dyd = dev(raw), dev(sc), dev(sh), dev(dy)
    assert gsd.lib.gsd_conv3x3_wgrad_takes_pitched_dy(n, h, w, ci, co) == 1
    need = gsd.lib.gsd_conv3x3_wgrad_workspace(n, h, w, ci, co)
    ws = torch.zeros(need, device="cuda")
    a_src = gsd.src_array([gsd.make_src(rawd, scd, shd, relu=True)])
    outs = []
    for t in (dyd, pitched(dyd)):
        dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
        dy_src = gsd.make_src(t)
        gsd.check(gsd.lib.gsd_conv3x3_wgrad(a_src, 1, C.byref(dy_src), ci, co, dw.data_ptr(), ws.data_ptr(), need, n, h, w,
                                            gsd.stream_ptr()))
        outs.append(dw.clone())
    assert rel_l1(outs[1].cpu().numpy(), dwr) < 5e-5
    assert torch.equal(outs[0], outs[1])
    # the direct form (first layer) refuses a pitched dy instead of mis-reading it
    assert gsd.lib.gsd_conv3x3_wgrad_takes_pitched_dy(n, h, w, 3, co) == 0
    x3 = torch.zeros((n, 3, h, w), device="cuda")
    need3 = gsd.lib.gsd_conv3x3_wgrad_workspace(n, h, w, 3, co)
    ws3 = torch.zeros(need3, device="cuda")
    dw3 = torch.zeros((co, 3, 3, 3), device="cuda")
    dyp = gsd.make_src(pitched(dyd))
    if w % 4:
        rc = gsd.lib.gsd_conv3x3_wgrad(gsd.src_array([gsd.make_src(x3)]), 1, C.byref(dyp), 3, co, dw3.data_ptr(), ws3.data_ptr(),
                                       need3, n, h, w, gsd.stream_ptr())
        assert rc == -2 and b"row-contiguous" in gsd.lib.gsd_last_error()


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 16, 24, 9, 11), (1, 64, 64, 21, 29), (1, 32, 64, 12, 427), (2, 128, 64, 17, 53)])
def test_conv3x3_w43_pitched_source_and_destination(gsd, n, ci, co, h, w):
    """The Winograd conv kernel with row-pitched operands: a plain source whose pitch padding holds 0 (halo windows move as
    aligned 16-byte pieces) and a pitched destination -- equal to the contiguous launch; the destination's padding columns are
    not touched."""
    rng = np.random.default_rng(ci + co + w)
    x = dev(rnd(rng, n, ci, h, w))
    wt_ = dev(rnd(rng, co, ci, 3, 3, scale=0.2))
    wl = layout(gsd, 4, wt_, co, ci)
    y0 = torch.zeros((n, co, h, w), device="cuda")
    gsd.check(gsd.lib.gsd_conv3x3_w43(gsd.src_array([gsd.make_src(x)]), 1, wl.data_ptr(), ci, co, gsd.dst_array([gsd.make_dst(y0)]), 1,
                                      None, n, h, w, gsd.stream_ptr()))
    pitch = (w + 3) // 4 * 4
    ybase = torch.full((n, co, h, pitch), 7.0, device="cuda")
    y1 = ybase[..., :w]
    gsd.check(gsd.lib.gsd_conv3x3_w43(gsd.src_array([gsd.make_src(pitched(x))]), 1, wl.data_ptr(), ci, co,
                                      gsd.dst_array([gsd.make_dst(y1)]), 1, None, n, h, w, gsd.stream_ptr()))
    assert torch.equal(y0, y1)
    assert bool((ybase[..., w:] == 7.0).all())


@pytest.mark.parametrize("n,ci,co,h,w", [(3, 16, 24, 9, 11), (5, 64, 64, 20, 26), (4, 32, 70, 13, 53), (2, 128, 64, 40, 53)])
def test_conv3x3_w43_row_folding(gsd, monkeypatch, n, ci, co, h, w):
    """Row folding (small images: the tile rows run over the padded flat rows of the whole batch, a block may cover the tail of
    one image and the head of the next; tiles 28 / 56 pixels wide): every output pixel is the same sum in the same order,
    so the results are bit-identical to the unfolded launch -- plain forward with deferred BatchNorm, two source segments
    with an F.pad offset, two cropped destinations, and the fused BatchNorm-backward dX epilogue (partial sums to rounding)."""
    rng = np.random.default_rng(n * 100 + w)
    raw, g_, b_, mean, invstd, scale, shift, a = _bn_setup(rng, n, ci, h, w)
    x, scd, shd = dev(raw), dev(scale), dev(shift)
    wt_ = dev(rnd(rng, co, ci, 3, 3, scale=0.2))
    wl = layout(gsd, 4, wt_, co, ci)
    c1 = max(4, co // 3) // 4 * 4
    up = dev(rnd(rng, n, c1, h - 2, w - 3))
    wt2 = dev(rnd(rng, co, ci + c1, 3, 3, scale=0.2))
    wl2 = layout(gsd, 4, wt2, co, ci + c1)
    wl2d = layout(gsd, 5, wt2, co, ci + c1)
    dy = dev(rnd(rng, n, co, h, w))
    wld = layout(gsd, 5, wt_, co, ci)
    vecs = [dev(v) for v in (scale, shift, mean, invstd)]
    res = {}
    for fold in ("0", "1"):
        monkeypatch.setenv("GSD_W43_FOLD", fold)
        rows = gsd.lib.gsd_conv3x3_w43_partial_rows(n, h, w, co)
        mpad = (co + 63) // 64 * 64
        part = torch.zeros(rows * 2 * mpad, device="cuda")
        y = torch.full((n, co, h, w), float("nan"), device="cuda")
        gsd.check(gsd.lib.gsd_conv3x3_w43(gsd.src_array([gsd.make_src(x, scd, shd, relu=True)]), 1, wl.data_ptr(), ci, co,
                                          gsd.dst_array([gsd.make_dst(y)]), 1, part.data_ptr(), n, h, w, gsd.stream_ptr()))
        sums = torch.zeros(65 * 2 * co, device="cuda", dtype=torch.float64)
        gsd.check(gsd.lib.gsd_bn_reduce_partials(part.data_ptr(), rows, mpad, co, sums.data_ptr(), gsd.stream_ptr()))
        y2 = torch.full((n, co, h, w), float("nan"), device="cuda")
        src2 = gsd.src_array([gsd.make_src(x, scd, shd, relu=True), gsd.make_src(up, off=(1, 1))])
        gsd.check(gsd.lib.gsd_conv3x3_w43(src2, 2, wl2.data_ptr(), ci + c1, co, gsd.dst_array([gsd.make_dst(y2)]), 1, None, n, h, w,
                                          gsd.stream_ptr()))
        g0 = torch.full((n, ci, h, w), float("nan"), device="cuda")
        g1 = torch.full((n, c1, h - 2, w - 3), float("nan"), device="cuda")
        gsd.check(gsd.lib.gsd_conv3x3_w43(gsd.src_array([gsd.make_src(dy)]), 1, wl2d.data_ptr(), co, ci + c1,
                                          gsd.dst_array([gsd.make_dst(g0), gsd.make_dst(g1, off=(1, 1))]), 2, None, n, h, w,
                                          gsd.stream_ptr()))
        rows_d = gsd.lib.gsd_conv3x3_w43_partial_rows(n, h, w, ci)
        mpad_d = (ci + 63) // 64 * 64
        part_d = torch.zeros(rows_d * 2 * mpad_d, device="cuda")
        dz = torch.full((n, ci, h, w), float("nan"), device="cuda")
        s, d = gsd.make_src(dy), gsd.make_dst(dz)
        gsd.check(gsd.lib.gsd_conv3x3_w43_dgrad_bnrelu(C.byref(s), wld.data_ptr(), co, ci, C.byref(d), x.data_ptr(),
                                                       *[v.data_ptr() for v in vecs], part_d.data_ptr(), n, h, w, gsd.stream_ptr()))
        sums_d = torch.zeros(65 * 2 * ci, device="cuda", dtype=torch.float64)
        gsd.check(gsd.lib.gsd_bn_reduce_partials(part_d.data_ptr(), rows_d, mpad_d, ci, sums_d.data_ptr(), gsd.stream_ptr()))
        res[fold] = (y, y2, g0, g1, dz, sums[:2 * co].clone(), sums_d[:2 * ci].clone(), rows)
    for k in range(5):
        assert bool(torch.isfinite(res["1"][k]).all()), k
        assert torch.equal(res["0"][k], res["1"][k]), k
    for k in (5, 6):
        np.testing.assert_allclose(res["1"][k].cpu().numpy(), res["0"][k].cpu().numpy(), rtol=1e-5, atol=1e-3)
    assert res["1"][7] <= res["0"][7]     # never more blocks (fewer at batch sizes that leave tail rows)


@pytest.mark.parametrize("n,ci,co,h,w,bn", [(2, 3, 64, 9, 37, True), (1, 3, 20, 5, 16, True), (3, 2, 70, 7, 50, True),
                                            (2, 3, 64, 6, 33, False), (5, 1, 16, 4, 3, True), (2, 3, 64, 41, 427, True)])
def test_conv3x3_wgrad_bn_first_layer(gsd, n, ci, co, h, w, bn):
    """dW of the first conv with the BatchNorm backward of its output applied on the fly (unet.py:15-16 in backward):
    same numbers as gsd_bn_bwd_apply's expression followed by the oracle's dW; twice the same bits; workspace checked."""
    from oracle import unet_numpy as on
    rng = np.random.default_rng(ci * 17 + co + w)
    x = rnd(rng, n, ci, h, w)
    dz, raw = rnd(rng, n, co, h, w), rnd(rng, n, co, h, w)
    sc, mu = rng.uniform(0.5, 1.5, co).astype(np.float32), rnd(rng, co, scale=0.3)
    istd, k1, k2 = rng.uniform(0.5, 2.0, co).astype(np.float32), rnd(rng, co, scale=0.1), rnd(rng, co, scale=0.1)
    b = lambda v: v[None, :, None, None]
    d_raw = (b(sc) * (dz - b(k1) - (raw - b(mu)) * b(istd) * b(k2))).astype(np.float32) if bn else dz
    _, dwr = on.conv3x3_bwd(x, np.zeros((co, ci, 3, 3), np.float32), d_raw, need_dx=False)
    assert gsd.lib.gsd_conv3x3_wgrad_bn_supported(n, h, w, ci, co) == 1
    assert gsd.lib.gsd_conv3x3_wgrad_bn_supported(n, h, w, 4, co) == 0
    need = gsd.lib.gsd_conv3x3_wgrad_bn_workspace(n, h, w, ci, co)
    ws = torch.zeros(need, device="cuda")
    xd, dzd, rawd = dev(x), dev(dz), dev(raw)
    par = [dev(v) for v in (sc, mu, istd, k1, k2)]
    ptrs = [p.data_ptr() for p in par] if bn else [None] * 5
    a_src = gsd.make_src(xd)
    outs = []
    for _ in range(2):
        dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
        gsd.check(gsd.lib.gsd_conv3x3_wgrad_bn(C.byref(a_src), dzd.data_ptr(), rawd.data_ptr() if bn else None, *ptrs, ci, co,
                                               dw.data_ptr(), ws.data_ptr(), need, n, h, w, gsd.stream_ptr()))
        outs.append(dw.cpu().numpy())
    assert np.isfinite(outs[0]).all()
    assert rel_l1(outs[0], dwr) < 5e-5
    assert np.array_equal(outs[0], outs[1])
    rc = gsd.lib.gsd_conv3x3_wgrad_bn(C.byref(a_src), dzd.data_ptr(), rawd.data_ptr() if bn else None, *ptrs, ci, co,
                                      dw.data_ptr(), ws.data_ptr(), need - 1, n, h, w, gsd.stream_ptr())
    assert rc == -4 and b"workspace" in gsd.lib.gsd_last_error()
    aff = gsd.make_src(xd, dev(np.ones(ci, np.float32)), dev(np.zeros(ci, np.float32)), relu=True)
    rc = gsd.lib.gsd_conv3x3_wgrad_bn(C.byref(aff), dzd.data_ptr(), rawd.data_ptr() if bn else None, *ptrs, ci, co,
                                      dw.data_ptr(), ws.data_ptr(), need, n, h, w, gsd.stream_ptr())
    assert rc != 0 and b"plain" in gsd.lib.gsd_last_error()


def with_slack(t, poison=float("nan")):
    """Copy of a contiguous device tensor with 4 POISONED floats in front of and behind it (what gsd_src.slack = 4 promises to
    be readable: a value that leaked from there into a result would show)."""
    buf = torch.full((t.numel() + 8,), poison, device=t.device, dtype=t.dtype)
    v = buf[4:4 + t.numel()].view(*t.shape)
    v.copy_(t)
    return v


@pytest.mark.parametrize("n,c0,c1,co,h,w,uh,uw", [(2, 32, 0, 64, 9, 37, 0, 0), (1, 64, 0, 128, 21, 53, 0, 0),
                                                  (2, 64, 64, 64, 13, 26, 12, 26), (1, 32, 32, 32, 10, 106, 10, 105),
                                                  (3, 16, 0, 16, 5, 16, 0, 0), (2, 128, 0, 64, 8, 213, 0, 0),
                                                  (1, 64, 64, 128, 7, 427, 6, 426), (1, 48, 0, 80, 6, 19, 0, 0)])
def test_conv3x3_wgrad_window_pieces(gsd, monkeypatch, n, c0, c1, co, h, w, uh, uw):
    """Activation windows as 16-byte pieces from unaligned rows (gsd_src.slack >= 4): image widths of every residue mod 4,
    images narrower than a tile row, two segments with an F.pad offset -- in the row form (GSD_WGRAD_W2D=0) the same bits as the
    dword-gather form (same products in the same order) and the oracle's numbers; the slack floats are NaN and must not leak.
    With slack the call may take the two-dimensional form (gsd_wgrad_w2d.hip, other products): there the oracle's numbers
    and no leak."""
    from oracle import unet_numpy as on
    rng = np.random.default_rng(c0 * 3 + c1 + w)
    raw0 = rnd(rng, n, c0, h, w)
    sc, sh = rng.uniform(0.5, 1.5, c0).astype(np.float32), rnd(rng, c0, scale=0.3)
    a = np.maximum(raw0 * sc[None, :, None, None] + sh[None, :, None, None], 0)
    r0d, r0s, scd, shd = dev(raw0), with_slack(dev(raw0)), dev(sc), dev(sh)
    segs = [gsd.make_src(r0d, scd, shd, relu=True)]
    keep = [r0d]
    segs_s = [gsd.make_src(r0s, scd, shd, relu=True, slack=4)]
    if c1:
        up = rnd(rng, n, c1, uh, uw)
        upp, (top, left) = on.pad_to(up, h, w)
        a = np.concatenate([a, upp], 1)
        upd, ups = dev(up), with_slack(dev(up))
        segs.append(gsd.make_src(upd, off=(top, left)))
        segs_s.append(gsd.make_src(ups, off=(top, left), slack=4))
        keep += [upd, ups]
    ci = c0 + c1
    dy = rnd(rng, n, co, h, w)
    _, dwr = on.conv3x3_bwd(a, np.zeros((co, ci, 3, 3), np.float32), dy, need_dx=False)
    dyp = pitched(dev(dy))
    dy_src = gsd.make_src(dyp)
    need = gsd.lib.gsd_conv3x3_wgrad_workspace(n, h, w, ci, co)
    ws = torch.zeros(need, device="cuda")
    def run(ss):
        dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
        gsd.check(gsd.lib.gsd_conv3x3_wgrad(gsd.src_array(ss), len(ss), C.byref(dy_src), ci, co, dw.data_ptr(), ws.data_ptr(),
                                            need, n, h, w, gsd.stream_ptr()))
        return dw.cpu().numpy()
    got = run(segs_s)                     # whichever form the library takes for these operands
    assert np.isfinite(got).all()
    assert rel_l1(got, dwr) < 5e-5
    monkeypatch.setenv("GSD_WGRAD_W2D", "0")
    outs = [run(segs), run(segs_s)]
    assert np.isfinite(outs[1]).all()
    assert rel_l1(outs[1], dwr) < 5e-5
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 16, 24, 9, 37), (3, 64, 64, 20, 26), (2, 32, 70, 13, 53), (1, 128, 64, 40, 106),
                                         (1, 64, 64, 11, 213), (1, 32, 64, 9, 427), (2, 16, 16, 5, 64), (4, 16, 32, 6, 7)])
def test_conv3x3_w43_unaligned_halo_pieces(gsd, monkeypatch, n, ci, co, h, w):
    """Halo windows as 16-byte pieces straight from unaligned rows (gsd_src.slack >= 4; GSD_W43_U4): image widths of every
    residue mod 4, images narrower than a tile (row folding), a deferred-BatchNorm source (NaN padding), a plain one, two
    segments with an F.pad offset and two cropped destinations, the fused BatchNorm-backward dX epilogue -- every launch
    bit-identical to the dword-gather form on the same data; the slack floats are NaN and must not leak."""
    monkeypatch.setenv("GSD_W43_U4", "2")   # every shape (by default only where most blocks lie inside the image)
    rng = np.random.default_rng(n * 100 + w + ci)
    raw, g_, b_, mean, invstd, scale, shift, a = _bn_setup(rng, n, ci, h, w)
    x, scd, shd = dev(raw), dev(scale), dev(shift)
    xs = with_slack(x)
    wt_ = dev(rnd(rng, co, ci, 3, 3, scale=0.2))
    wl = layout(gsd, 4, wt_, co, ci)
    c1 = max(4, co // 3) // 4 * 4
    up = dev(rnd(rng, n, c1, h - 2, w - 3))
    ups = with_slack(up)
    wt2 = dev(rnd(rng, co, ci + c1, 3, 3, scale=0.2))
    wl2 = layout(gsd, 4, wt2, co, ci + c1)
    dy = dev(rnd(rng, n, co, h, w))
    dys = with_slack(dy)
    wld = layout(gsd, 5, wt_, co, ci)
    vecs = [dev(v) for v in (scale, shift, mean, invstd)]
    res = {}
    for tag, xx, uu, dd, sl in (("dword", x, up, dy, 0), ("pieces", xs, ups, dys, 4)):
        y = torch.full((n, co, h, w), float("nan"), device="cuda")
        gsd.check(gsd.lib.gsd_conv3x3_w43(gsd.src_array([gsd.make_src(xx, scd, shd, relu=True, slack=sl)]), 1, wl.data_ptr(), ci,
                                          co, gsd.dst_array([gsd.make_dst(y)]), 1, None, n, h, w, gsd.stream_ptr()))
        y1 = torch.full((n, co, h, w), float("nan"), device="cuda")
        gsd.check(gsd.lib.gsd_conv3x3_w43(gsd.src_array([gsd.make_src(xx, slack=sl)]), 1, wl.data_ptr(), ci, co,
                                          gsd.dst_array([gsd.make_dst(y1)]), 1, None, n, h, w, gsd.stream_ptr()))
        y2 = torch.full((n, co, h, w), float("nan"), device="cuda")
        gsd.check(gsd.lib.gsd_conv3x3_w43(gsd.src_array([gsd.make_src(xx, scd, shd, relu=True, slack=sl),
                                                         gsd.make_src(uu, off=(1, 2), slack=sl)]), 2, wl2.data_ptr(), ci + c1, co,
                                          gsd.dst_array([gsd.make_dst(y2)]), 1, None, n, h, w, gsd.stream_ptr()))
        rows = gsd.lib.gsd_conv3x3_w43_partial_rows(n, h, w, ci)
        mpad = (ci + 63) // 64 * 64
        part = torch.zeros(rows * 2 * mpad, device="cuda")
        dz = torch.full((n, ci, h, w), float("nan"), device="cuda")
        sdy = gsd.make_src(dd, slack=sl)
        gsd.check(gsd.lib.gsd_conv3x3_w43_dgrad_bnrelu(C.byref(sdy), wld.data_ptr(), co, ci, C.byref(gsd.make_dst(dz)), x.data_ptr(),
                                                       *[v.data_ptr() for v in vecs], part.data_ptr(), n, h, w, gsd.stream_ptr()))
        torch.cuda.synchronize()
        res[tag] = (y, y1, y2, dz, part)
    for a_, b_ in zip(res["dword"], res["pieces"]):
        assert bool(torch.isfinite(b_).all())
        assert torch.equal(a_, b_)
    # and the numbers themselves: against the direct-tap kernel
    wl0 = layout(gsd, 0, wt_, co, ci)
    yd = torch.full((n, co, h, w), float("nan"), device="cuda")
    gsd.check(gsd.lib.gsd_conv3x3(gsd.src_array([gsd.make_src(x, scd, shd, relu=True)]), 1, wl0.data_ptr(), ci, co,
                                  gsd.dst_array([gsd.make_dst(yd)]), 1, None, n, h, w, gsd.stream_ptr()))
    assert rel_l1(res["pieces"][0].cpu().numpy(), yd.cpu().numpy()) < TOL


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 32, 64, 12, 37), (1, 64, 128, 8, 64), (2, 128, 64, 9, 50)])
def test_winograd_instantiation_switches_are_bit_identical(gsd, monkeypatch, n, ci, co, h, w):
    """The tuning switches only choose between instantiations that multiply the same numbers in the same order: window-row
    reuse in the dW kernel (GSD_WG43_RR), plain-source transforms in dW and conv (GSD_WG43_PLAIN / GSD_W43_PLAIN)."""
    rng = np.random.default_rng(ci + co + w)
    x = with_slack(dev(rnd(rng, n, ci, h, w)), poison=0.0)
    sc, sh = dev(rng.uniform(0.5, 1.5, ci).astype(np.float32)), dev(rnd(rng, ci, scale=0.3))
    dy = pitched(dev(rnd(rng, n, co, h, w)))
    dy_src = gsd.make_src(dy)
    need = gsd.lib.gsd_conv3x3_wgrad_workspace(n, h, w, ci, co)
    ws = torch.zeros(need, device="cuda")
    wl = layout(gsd, 4, dev(rnd(rng, co, ci, 3, 3, scale=0.2)), co, ci)

    def dw_of(src):
        dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
        gsd.check(gsd.lib.gsd_conv3x3_wgrad(gsd.src_array([src]), 1, C.byref(dy_src), ci, co, dw.data_ptr(), ws.data_ptr(), need,
                                            n, h, w, gsd.stream_ptr()))
        return dw

    def conv_of(src):
        y = torch.full((n, co, h, w), float("nan"), device="cuda")
        gsd.check(gsd.lib.gsd_conv3x3_w43(gsd.src_array([src]), 1, wl.data_ptr(), ci, co, gsd.dst_array([gsd.make_dst(y)]), 1, None,
                                          n, h, w, gsd.stream_ptr()))
        return y

    bn_src, plain_src = gsd.make_src(x, sc, sh, relu=True, slack=4), gsd.make_src(x, slack=4)
    ref = dw_of(bn_src), dw_of(plain_src), conv_of(plain_src)
    for var in ("GSD_WG43_RR", "GSD_WG43_PLAIN", "GSD_W43_PLAIN"):
        monkeypatch.setenv(var, "0")
        got = dw_of(bn_src), dw_of(plain_src), conv_of(plain_src)
        monkeypatch.delenv(var)
        for a_, b_ in zip(ref, got):
            assert bool(torch.isfinite(a_).all()) and torch.equal(a_, b_), var


@pytest.mark.parametrize("n,ci,co,h,w,mg,pg", [(2, 512, 256, 13, 53, None, None), (3, 64, 320, 20, 37, 3, 5), (2, 32, 448, 9, 70, 2, 1),
                                               (1, 128, 192, 17, 33, 1, 64)])
def test_w2d_block_order_is_bit_identical(gsd, monkeypatch, n, ci, co, h, w, mg, pg):
    """The deep levels walk their blocks in passes of a few m-blocks over groups of pixel tiles (so that the weight images in flight
    fit the L2; gsd_conv3x3_w2d.hip, GSD_W2D_MGROUP / GSD_W2D_PGROUP): a permutation of the block ids -- every output element and
    every BatchNorm partial row is written exactly once (NaN-prefilled), with the bits of the plain order.  Forced group sizes that
    leave a short last pixel group and a short last pass; the default rule at K = 512."""
    rng = np.random.default_rng(ci + co)
    x = with_slack(dev(rnd(rng, n, ci, h, w)))
    sc, sh = dev(rng.uniform(0.5, 1.5, ci).astype(np.float32)), dev(rnd(rng, ci, scale=0.3))
    wl = layout(gsd, 8, dev(rnd(rng, co, ci, 3, 3, scale=0.1)), co, ci)
    rows = gsd.lib.gsd_conv3x3_w2d_partial_rows(n, h, w, co)
    mpad = (co + 63) // 64 * 64
    res = []
    for order in ("plain", "grouped"):
        if order == "plain":
            monkeypatch.setenv("GSD_W2D_MGROUP", "0")
        elif mg is None:
            monkeypatch.delenv("GSD_W2D_MGROUP")
        else:
            monkeypatch.setenv("GSD_W2D_MGROUP", str(mg))
            monkeypatch.setenv("GSD_W2D_PGROUP", str(pg))
        y = torch.full((n, co, h, w), float("nan"), device="cuda")
        part = torch.full((rows * 2 * mpad,), float("nan"), device="cuda")
        gsd.check(gsd.lib.gsd_conv3x3_w2d(gsd.src_array([gsd.make_src(x, sc, sh, relu=True, slack=4)]), 1, wl.data_ptr(), ci, co,
                                          gsd.dst_array([gsd.make_dst(y)]), 1, part.data_ptr(), n, h, w, gsd.stream_ptr()))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(part).all()), order
        res.append((y, part))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
