"""GPU parity at the REAL layer shapes of BASELINE.json's network ([64,128,256,512,1024] @ 3x320x427; reference
gelslim_depth/models/unet.py:67-77): every conv3x3 unit and every transposed conv of the U-Net, launched the way the engine
launches it (deferred BatchNorm+ReLU sources with slack, the two-segment decoder form with its F.pad offset, row-pitched dy,
the fused BatchNorm-backward dX epilogue, two cropped dX destinations), against oracle/unet_numpy.py at the op tolerances.

Tile choosers, split-K block counts, row folding and the XCD swizzle depend on (N, H, W, Cin, Cout): the small odd shapes of
tests/test_gpu_ops.py do not reach the instantiations the benchmark runs.  Batch 2 everywhere; batch 32 in addition at the
40x53 / 20x26 levels, where row folding spans images (forward and dX are checked on three images of the 32 -- images are
independent -- dW on all of them, accumulated in fp64 over chunks of the oracle's fp32 BLAS sums).
"""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch

from conftest import rel_l1

pytestmark = pytest.mark.gpu

DIMS = [64, 128, 256, 512, 1024]
HS = [320, 160, 80, 40, 20]
WS = [427, 213, 106, 53, 26]

# (name, level, skip/plain channels, up-sampled channels (two-segment decoder form), Cout, input is a pooled (plain) tensor)
UNITS = [
    ("inc.c1|up3.c1", 0, 64, 0, 64, False),
    ("down0.c0", 1, 64, 0, 128, True),
    ("down0.c1|up2.c1", 1, 128, 0, 128, False),
    ("down1.c0", 2, 128, 0, 256, True),
    ("down1.c1|up1.c1", 2, 256, 0, 256, False),
    ("down2.c0", 3, 256, 0, 512, True),
    ("down2.c1|up0.c1", 3, 512, 0, 512, False),
    ("down3.c0", 4, 512, 0, 1024, True),
    ("down3.c1", 4, 1024, 0, 1024, False),
    ("up0.c0", 3, 512, 512, 512, False),
    ("up1.c0", 2, 256, 256, 256, False),
    ("up2.c0", 1, 128, 128, 128, False),
    ("up3.c0", 0, 64, 64, 64, False),
]
CASES = [(u, 2) for u in UNITS] + [(u, 32) for u in UNITS if u[1] >= 3]


@pytest.fixture(scope="module")
def gsd():
    from gelslim_depth_amd import _lib
    return _lib


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def slack_dev(gsd, a):
    """Device copy of `a` the way the engine allocates activations: 4 readable floats either side (gsd_src.slack)."""
    t = gsd.slack_empty(a.shape, "cuda")
    t.copy_(torch.from_numpy(np.ascontiguousarray(a)))
    return t


def pitched(t):
    n, c, h, w = t.shape
    base = torch.zeros((n, c, h, (w + 3) // 4 * 4), device=t.device, dtype=t.dtype)
    base[..., :w] = t
    return base[..., :w]


def rnd(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape, dtype=np.float32) * np.float32(scale))


def layout(gsd, mode, w, co, ci):
    wt = torch.zeros(gsd.lib.gsd_weight_layout_size(mode, co, ci), device="cuda")
    gsd.check(gsd.lib.gsd_weight_layout(mode, w.data_ptr(), co, ci, wt.data_ptr(), gsd.stream_ptr()))
    return wt


def bcast(v):
    return v[None, :, None, None]


FORM_CASES = [(u, n, f) for u, n in CASES for f in ("w43", "w2d")]


@pytest.mark.parametrize("unit,n,form", FORM_CASES, ids=[f"{u[0]}-N{n}-{f}" for u, n, f in FORM_CASES])
def test_conv3x3_unit_at_network_shape(gsd, unit, n, form):
    """form: the Winograd F(4,3)-rows kernels (gsd_conv3x3_w43*) or the two-dimensional F(2x4,3x3) kernels (gsd_conv3x3_w2d*) for
    the forward and dX launches; dW is the same kernel in both (the two-dimensional Winograd form, checked in the w43 case)."""
    from oracle import unet_numpy as on
    name, lvl, c0, c1, co, pooled = unit
    h, w = HS[lvl], WS[lvl]
    ci = c0 + c1
    L = gsd.lib
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 10000 + n)
    assert L.gsd_conv3x3_algo(n, h, w, ci, co) == 1 and L.gsd_conv3x3_algo(n, h, w, co, ci) == 1, "Winograd layers"
    if form == "w2d":
        assert L.gsd_conv3x3_w2d_supported(ci, c0) == 1 and L.gsd_conv3x3_w2d_supported(co, co) == 1
        conv, conv_bn, conv_rows, mode_f, mode_d = (L.gsd_conv3x3_w2d, L.gsd_conv3x3_w2d_dgrad_bnrelu, L.gsd_conv3x3_w2d_partial_rows, 8, 9)
    else:
        conv, conv_bn, conv_rows, mode_f, mode_d = (L.gsd_conv3x3_w43, L.gsd_conv3x3_w43_dgrad_bnrelu, L.gsd_conv3x3_w43_partial_rows, 4, 5)
    sub = list(range(n)) if n <= 2 else [0, 13, n - 1]          # images whose forward / dX the oracle computes

    # ---- operands as the engine holds them
    raw0 = rnd(rng, n, c0, h, w)
    if pooled:
        a0 = raw0
        r0d = slack_dev(gsd, raw0)
        segs = [gsd.make_src(r0d, slack=gsd.SLACK)]
        keep = [r0d]
    else:
        sc, sh = rng.uniform(0.5, 1.5, c0).astype(np.float32), rnd(rng, c0, scale=0.3)
        a0 = np.maximum(raw0 * bcast(sc) + bcast(sh), 0)
        r0d, scd, shd = slack_dev(gsd, raw0), dev(sc), dev(sh)
        segs = [gsd.make_src(r0d, scd, shd, relu=True, slack=gsd.SLACK)]
        keep = [r0d, scd, shd]
    top = left = 0
    if c1:
        uh, uw = 2 * HS[lvl + 1], 2 * WS[lvl + 1]
        up = rnd(rng, n, c1, uh, uw)
        upp, (top, left) = on.pad_to(up, h, w)
        a = np.concatenate([a0, upp], 1)
        upd = slack_dev(gsd, up)
        segs.append(gsd.make_src(upd, off=(top, left), slack=gsd.SLACK))
        keep.append(upd)
        del upp
    else:
        a = a0
    wt_ = rnd(rng, co, ci, 3, 3, scale=1.0 / np.sqrt(9 * ci))
    wd = dev(wt_)
    src = gsd.src_array(segs)

    # ---- forward (+ BatchNorm partial sums)
    y = torch.full((n, co, h, w), float("nan"), device="cuda")
    rows = conv_rows(n, h, w, co)
    mpad = (co + 63) // 64 * 64
    part = torch.zeros(rows * 2 * mpad, device="cuda")
    gsd.check(conv(src, len(segs), layout(gsd, mode_f, wd, co, ci).data_ptr(), ci, co, gsd.dst_array([gsd.make_dst(y)]), 1,
                   part.data_ptr(), n, h, w, gsd.stream_ptr()))
    assert bool(torch.isfinite(y).all()), "every output element must be written"
    ref = on.conv3x3_fwd(a[sub], wt_)
    assert rel_l1(y[sub].cpu().numpy(), ref) < 1e-5
    sums = torch.zeros(65 * 2 * co, device="cuda", dtype=torch.float64)
    gsd.check(L.gsd_bn_reduce_partials(part.data_ptr(), rows, mpad, co, sums.data_ptr(), gsd.stream_ptr()))
    y64 = y.double()
    np.testing.assert_allclose(sums[:co].cpu().numpy(), y64.sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(sums[co:2 * co].cpu().numpy(), (y64 * y64).sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-5)
    del y, y64, ref

    # ---- dX, in the form the engine launches for this unit
    dy = rnd(rng, n, co, h, w)
    dyp = pitched(dev(dy))
    wl_d = layout(gsd, mode_d, wd, co, ci)
    dxr, _ = on.conv3x3_bwd(a[sub], wt_, dy[sub])
    if c1:        # decoder c0: (skip gradient | cropped gradient of the up-sampled tensor) + its per-channel sums (ConvT bias grad)
        g_skip = torch.full((n, c0, h, w), float("nan"), device="cuda")
        g_up = torch.full((n, c1, uh, uw), float("nan"), device="cuda")
        rows_d = conv_rows(n, h, w, ci)
        part_d = torch.zeros(rows_d * 2 * ((ci + 63) // 64 * 64), device="cuda")
        gsd.check(conv(gsd.src_array([gsd.make_src(dyp)]), 1, wl_d.data_ptr(), co, ci,
                       gsd.dst_array([gsd.make_dst(g_skip), gsd.make_dst(g_up, off=(top, left))]), 2,
                       part_d.data_ptr(), n, h, w, gsd.stream_ptr()))
        assert rel_l1(g_skip[sub].cpu().numpy(), dxr[:, :c0]) < 1e-5
        assert rel_l1(g_up[sub].cpu().numpy(), dxr[:, c0:, top:top + uh, left:left + uw]) < 1e-5
        sums_d = torch.zeros(65 * 2 * ci, device="cuda", dtype=torch.float64)
        gsd.check(L.gsd_bn_reduce_partials(part_d.data_ptr(), rows_d, (ci + 63) // 64 * 64, ci, sums_d.data_ptr(), gsd.stream_ptr()))
        np.testing.assert_allclose(sums_d[c0:ci].cpu().numpy(), g_up.double().sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-4, atol=1e-2)
        del g_skip, g_up
    elif pooled:  # encoder c0 below level 0: plain dX into the pooled tensor's gradient
        g = torch.full((n, ci, h, w), float("nan"), device="cuda")
        gsd.check(conv(gsd.src_array([gsd.make_src(dyp)]), 1, wl_d.data_ptr(), co, ci,
                       gsd.dst_array([gsd.make_dst(g)]), 1, None, n, h, w, gsd.stream_ptr()))
        assert rel_l1(g[sub].cpu().numpy(), dxr) < 1e-5
        del g
    else:         # c1 of a DoubleConv: dX fused with the backward of the producer's ReLU + BatchNorm reduce pass
        mean, invstd = rnd(rng, c0, scale=0.3), rng.uniform(0.5, 2.0, c0).astype(np.float32)
        dz_ref = dxr * (a0[sub] > 0)
        vecs = [scd, shd, dev(mean), dev(invstd)]
        dz = torch.full((n, ci, h, w), float("nan"), device="cuda")
        rows_d = conv_rows(n, h, w, ci)
        mp = (ci + 63) // 64 * 64
        part_d = torch.zeros(rows_d * 2 * mp, device="cuda")
        s, d = gsd.make_src(dyp), gsd.make_dst(dz)
        gsd.check(conv_bn(C.byref(s), wl_d.data_ptr(), co, ci, C.byref(d), r0d.data_ptr(),
                          *[v.data_ptr() for v in vecs], part_d.data_ptr(), n, h, w, gsd.stream_ptr()))
        assert rel_l1(dz[sub].cpu().numpy(), dz_ref) < 1e-5
        sums_d = torch.zeros(65 * 2 * ci, device="cuda", dtype=torch.float64)
        gsd.check(L.gsd_bn_reduce_partials(part_d.data_ptr(), rows_d, mp, ci, sums_d.data_ptr(), gsd.stream_ptr()))
        dz64 = dz.double()
        xhat = (r0d.double() - vecs[2].double()[None, :, None, None]) * vecs[3].double()[None, :, None, None]
        np.testing.assert_allclose(sums_d[:ci].cpu().numpy(), dz64.sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-4, atol=1e-2)
        np.testing.assert_allclose(sums_d[ci:2 * ci].cpu().numpy(), (dz64 * xhat).sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-4,
                                   atol=1e-2)
        del dz, dz64, xhat
    del dxr

    if form == "w2d":
        return          # dW does not depend on the forward / dX form: checked once, in the w43 case
    # ---- dW (activation segments as in the forward, dy from the row-pitched buffer)
    dwr = np.zeros((co, ci, 3, 3), np.float64)
    wdummy = np.zeros((co, ci, 3, 3), np.float32)
    for i in range(0, n, 4):
        dwr += on.conv3x3_bwd(a[i:i + 4], wdummy, dy[i:i + 4], need_dx=False)[1]
    need = L.gsd_conv3x3_wgrad_workspace(n, h, w, ci, co)
    ws = torch.zeros(need, device="cuda")
    dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
    assert L.gsd_conv3x3_wgrad_takes_pitched_dy(n, h, w, ci, co) == 1
    dy_src = gsd.make_src(dyp)
    # every Winograd layer of the network takes the two-dimensional F(2x4,3x3) dW kernel (gsd_wgrad_w2d.hip) in the engine's form
    assert L.gsd_conv3x3_wgrad_form(src, len(segs), C.byref(dy_src), ci, co, n, h, w) == 2
    gsd.check(L.gsd_conv3x3_wgrad(src, len(segs), C.byref(dy_src), ci, co, dw.data_ptr(), ws.data_ptr(), need, n, h, w,
                                  gsd.stream_ptr()))
    got = dw.cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_l1(got, dwr) < 5e-5
    if c1:
        assert rel_l1(got[:, c0:], dwr[:, c0:]) < 5e-5      # the offset segment on its own
    del keep


@pytest.mark.parametrize("n", [2, 32])
def test_first_layer_at_network_shape(gsd, n):
    """inc.c0 (3 -> 64 @320x427): direct-tap forward with BatchNorm partial sums, and the dW kernel that applies the BatchNorm
    backward of its own output on the fly (no dX: the input is the image)."""
    from oracle import unet_numpy as on
    L = gsd.lib
    ci, co, h, w = 3, 64, HS[0], WS[0]
    rng = np.random.default_rng(n)
    assert L.gsd_conv3x3_algo(n, h, w, ci, co) == 0
    sub = list(range(n)) if n <= 2 else [0, 13, n - 1]
    x = rng.random((n, ci, h, w), dtype=np.float32)
    wt_ = rnd(rng, co, ci, 3, 3, scale=0.2)
    xd, wd = dev(x), dev(wt_)
    y = torch.full((n, co, h, w), float("nan"), device="cuda")
    rows = L.gsd_conv3x3_partial_rows(n, h, w, co)
    part = torch.zeros(rows * 2 * 64, device="cuda")
    gsd.check(L.gsd_conv3x3(gsd.src_array([gsd.make_src(xd)]), 1, layout(gsd, 0, wd, co, ci).data_ptr(), ci, co,
                            gsd.dst_array([gsd.make_dst(y)]), 1, part.data_ptr(), n, h, w, gsd.stream_ptr()))
    assert rel_l1(y[sub].cpu().numpy(), on.conv3x3_fwd(x[sub], wt_)) < 2e-5
    sums = torch.zeros(65 * 2 * co, device="cuda", dtype=torch.float64)
    gsd.check(L.gsd_bn_reduce_partials(part.data_ptr(), rows, 64, co, sums.data_ptr(), gsd.stream_ptr()))
    np.testing.assert_allclose(sums[:co].cpu().numpy(), y.double().sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=1e-2)
    # dW with the BatchNorm backward on the fly: d_raw = scale * (dz - k1 - (raw - mean) * invstd * k2)
    assert L.gsd_conv3x3_wgrad_bn_supported(n, h, w, ci, co) == 1
    dz = rnd(rng, n, co, h, w)
    raw = y.cpu().numpy()
    sc, mu = rng.uniform(0.5, 1.5, co).astype(np.float32), rnd(rng, co, scale=0.3)
    istd, k1, k2 = rng.uniform(0.5, 2.0, co).astype(np.float32), rnd(rng, co, scale=0.1), rnd(rng, co, scale=0.1)
    d_raw = (bcast(sc) * (dz - bcast(k1) - (raw - bcast(mu)) * bcast(istd) * bcast(k2))).astype(np.float32)
    dwr = np.zeros((co, ci, 3, 3), np.float64)
    for i in range(0, n, 4):
        dwr += on.conv3x3_bwd(x[i:i + 4], np.zeros((co, ci, 3, 3), np.float32), d_raw[i:i + 4], need_dx=False)[1]
    need = L.gsd_conv3x3_wgrad_bn_workspace(n, h, w, ci, co)
    ws = torch.zeros(need, device="cuda")
    par = [dev(v) for v in (sc, mu, istd, k1, k2)]
    dzd = dev(dz)
    dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
    a_src = gsd.make_src(xd)
    gsd.check(L.gsd_conv3x3_wgrad_bn(C.byref(a_src), dzd.data_ptr(), y.data_ptr(), *[p.data_ptr() for p in par], ci, co,
                                     dw.data_ptr(), ws.data_ptr(), need, n, h, w, gsd.stream_ptr()))
    assert rel_l1(dw.cpu().numpy(), dwr) < 5e-5


@pytest.mark.parametrize("lvl,c0,c1,co", [(3, 512, 512, 512), (1, 128, 0, 128), (0, 64, 64, 64)])
def test_w2d_halo_forms_are_bit_identical(gsd, monkeypatch, lvl, c0, c1, co):
    """The three ways gsd_conv3x3_w2d moves its halo windows -- dword gathers (default for unaligned sources), unaligned 16-byte
    pieces with the right-edge patch (GSD_W2D_U4=1) and, for a row-pitched plain source, aligned 16-byte pieces (default there;
    GSD_W2D_X4=0: dword gathers) -- bring the same values to the same products: outputs are bit-identical.  Two-segment source
    with the F.pad offset (segment switch + both segments' right edges), deferred BatchNorm, every tile column of the level."""
    from oracle import unet_numpy as on
    L = gsd.lib
    n, h, w = 2, HS[lvl], WS[lvl]
    ci = c0 + c1
    rng = np.random.default_rng(lvl + 5)
    sc, sh = dev(rng.uniform(0.5, 1.5, c0).astype(np.float32)), dev(rnd(rng, c0, scale=0.3))
    r0 = slack_dev(gsd, rnd(rng, n, c0, h, w))
    segs = [gsd.make_src(r0, sc, sh, relu=True, slack=gsd.SLACK)]
    keep = [r0]
    if c1:
        uh, uw = 2 * HS[lvl + 1], 2 * WS[lvl + 1]
        up = rnd(rng, n, c1, uh, uw)
        _, (top, left) = on.pad_to(up[:1, :1], h, w)
        upd = slack_dev(gsd, up)
        segs.append(gsd.make_src(upd, off=(top, left), slack=gsd.SLACK))
        keep.append(upd)
    wd = dev(rnd(rng, co, ci, 3, 3, scale=1.0 / np.sqrt(9 * ci)))
    wl_f, wl_d = layout(gsd, 8, wd, co, ci), layout(gsd, 9, wd, co, ci)
    src = gsd.src_array(segs)
    dyp = pitched(dev(rnd(rng, n, co, h, w)))
    outs = {}
    for tag, env in (("dword", {"GSD_W2D_U4": "0", "GSD_W2D_X4": "0"}), ("pieces", {"GSD_W2D_U4": "1", "GSD_W2D_X4": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        y = torch.full((n, co, h, w), float("nan"), device="cuda")
        gsd.check(L.gsd_conv3x3_w2d(src, len(segs), wl_f.data_ptr(), ci, co, gsd.dst_array([gsd.make_dst(y)]), 1, None, n, h, w,
                                    gsd.stream_ptr()))
        g = torch.full((n, ci, h, w), float("nan"), device="cuda")
        gsd.check(L.gsd_conv3x3_w2d(gsd.src_array([gsd.make_src(dyp)]), 1, wl_d.data_ptr(), co, ci, gsd.dst_array([gsd.make_dst(g)]), 1,
                                    None, n, h, w, gsd.stream_ptr()))
        outs[tag] = (y, g)
    assert bool(torch.isfinite(outs["pieces"][0]).all()) and bool(torch.isfinite(outs["pieces"][1]).all())
    assert torch.equal(outs["dword"][0], outs["pieces"][0]), "forward: unaligned pieces vs dword gathers"
    assert torch.equal(outs["dword"][1], outs["pieces"][1]), "dX: aligned pieces vs dword gathers"


CONVT = [("up0.up", 4, 1024), ("up1.up", 3, 512), ("up2.up", 2, 256), ("up3.up", 1, 128)]
CONVT_CASES = [(c, 2) for c in CONVT] + [(c, 32) for c in CONVT[:2]]


@pytest.mark.parametrize("case,n", CONVT_CASES, ids=[f"{c[0]}-N{n}" for c, n in CONVT_CASES])
def test_convT_at_network_shape(gsd, case, n):
    """ConvTranspose2d(Cin, Cin/2, 2, 2) of the four decoder levels (unet.py:36,41): forward (+bias) from a deferred
    BatchNorm+ReLU source, dX, dW and the bias gradient."""
    from oracle import unet_numpy as on
    name, lvl, ci = case
    L = gsd.lib
    co, h, w = ci // 2, HS[lvl], WS[lvl]
    rng = np.random.default_rng(ci + n)
    raw = rnd(rng, n, ci, h, w)
    sc, sh = rng.uniform(0.5, 1.5, ci).astype(np.float32), rnd(rng, ci, scale=0.3)
    x = np.maximum(raw * bcast(sc) + bcast(sh), 0)
    wt_, b = rnd(rng, ci, co, 2, 2, scale=1.0 / np.sqrt(ci)), rnd(rng, co)
    rawd, scd, shd, wd, bd = slack_dev(gsd, raw), dev(sc), dev(sh), dev(wt_), dev(b)
    y = torch.full((n, co, 2 * h, 2 * w), float("nan"), device="cuda")
    s, d = gsd.make_src(rawd, scd, shd, relu=True, slack=gsd.SLACK), gsd.make_dst(y)
    gsd.check(L.gsd_convT2x2(C.byref(s), layout(gsd, 6, wd, co, ci).data_ptr(), bd.data_ptr(), ci, co, C.byref(d), n, h, w,
                             gsd.stream_ptr()))
    assert rel_l1(y.cpu().numpy(), on.convT_fwd(x, wt_, b)) < 2e-5
    del y
    dy = rnd(rng, n, co, 2 * h, 2 * w)
    dxr, dwr, dbr = on.convT_bwd(x, wt_, dy)
    if n > 4:     # fp64 accumulation over chunks for the long reduction
        dwr = np.zeros((ci, co, 2, 2), np.float64)
        for i in range(0, n, 4):
            dwr += on.convT_bwd(x[i:i + 4], wt_, dy[i:i + 4])[1]
    dyd = slack_dev(gsd, dy)           # as the engine allocates up.dout
    dx = torch.full((n, ci, h, w), float("nan"), device="cuda")
    sdy, ddx = gsd.make_src(dyd, slack=gsd.SLACK), gsd.make_dst(dx)
    mode = L.gsd_convT2x2_dgrad_layout(C.byref(sdy), ci, co, n, h, w)
    assert mode == 7, "the LDS-DMA dX kernel runs at every level of the network"
    gsd.check(L.gsd_convT2x2_dgrad(C.byref(sdy), layout(gsd, mode, wd, co, ci).data_ptr(), ci, co, C.byref(ddx), n, h, w,
                                   gsd.stream_ptr()))
    assert bool(torch.isfinite(dx).all())
    assert rel_l1(dx.cpu().numpy(), dxr) < 2e-5
    need = L.gsd_convT2x2_wgrad_workspace(n, h, w, ci, co)
    ws = torch.zeros(need, device="cuda")
    dw = torch.full((ci, co, 2, 2), float("nan"), device="cuda")
    db = torch.full((co,), float("nan"), device="cuda")
    gsd.check(L.gsd_convT2x2_wgrad(C.byref(s), C.byref(sdy), ci, co, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), need, n, h, w,
                                   gsd.stream_ptr()))
    assert rel_l1(dw.cpu().numpy(), dwr) < 5e-5
    assert rel_l1(db.cpu().numpy(), dbr) < 2e-5
