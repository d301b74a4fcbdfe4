"""GPU parity of the bf16 kernel family at the REAL layer shapes of BASELINE.json's network ([64,128,256,512,1024] @320x427):
every conv3x3 unit as the bf16 engine launches it -- forward with BatchNorm partial sums, dX (plain, or with pass 1 of the
producer's BatchNorm+ReLU backward fused where the engine fuses it), dW -- and the four transposed convolutions, against fp64
evaluations of the same contractions on the same bf16-rounded operands (tests/test_gpu_bf16.py's bounds: one bf16 ulp for
stored bf16 results, 2e-5 of the largest element for fp32 weight gradients).

The persistent grids, the XCD-aware work order, the split-K block counts and the tile shapes all depend on (N, H, W, K, M):
the small shapes of tests/test_gpu_bf16.py do not reach what the benchmark runs (round 3: a partial-row index that was only
wrong for grids that are multiples of 8 with more than one m-block slipped past them).  Batch 2 everywhere; batch 32 in
addition at the 40x53 / 20x26 levels (forward and dX checked on three images, statistics and dW over all of them).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_bf16 import T3X, T3Y, _lib, _wgrad, assert_close_bf16, bf16r, from_nhwc, to_nhwc

pytestmark = pytest.mark.gpu

HS = [320, 160, 80, 40, 20]
WS = [427, 213, 106, 53, 26]
# (name, level, Cin, Cout, dX has the fused BatchNorm-backward epilogue, has a dX at all)
UNITS = [
    ("inc.c1|up3.c1", 0, 64, 64, True),
    ("down0.c0", 1, 64, 128, False),
    ("down0.c1|up2.c1", 1, 128, 128, True),
    ("down1.c0", 2, 128, 256, False),
    ("down1.c1|up1.c1", 2, 256, 256, True),
    ("down2.c0", 3, 256, 512, False),
    ("down2.c1|up0.c1", 3, 512, 512, True),
    ("down3.c0", 4, 512, 1024, False),
    ("down3.c1", 4, 1024, 1024, True),
    ("up0.c0", 3, 1024, 512, False),
    ("up1.c0", 2, 512, 256, False),
    ("up2.c0", 1, 256, 128, False),
    ("up3.c0", 0, 128, 64, False),
]
CASES = [(u, 2) for u in UNITS] + [(u, 32) for u in UNITS if u[1] >= 3]


def image(L, mode, w_d, co, ci):
    img = torch.empty((L.lib.gsd_bf16_weight_image_size(mode, co, ci),), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib.gsd_bf16_weight_image(mode, w_d.data_ptr(), co, ci, img.data_ptr(), L.stream_ptr()), "wimg")
    return img


@pytest.mark.parametrize("unit,n", CASES, ids=[f"{u[0]}-N{n}" for u, n in CASES])
def test_bf16_conv3x3_unit_at_network_shape(unit, n):
    L = _lib()
    name, lvl, cin, cout, fused = unit
    h, w = HS[lvl], WS[lvl]
    g = torch.Generator().manual_seed(lvl * 100 + cin + n)
    sub = list(range(n)) if n <= 2 else [0, 13, n - 1]
    x = bf16r(torch.randn((n, cin, h, w), generator=g))
    wt = torch.randn((cout, cin, 3, 3), generator=g) / (3.0 * cin ** 0.5)
    wq = bf16r(wt)
    w_d = wt.cuda()
    xin = to_nhwc(x)
    mp = L.lib.gsd_bf16_conv_mpad(cout)

    # ---- forward + BatchNorm partial sums
    out = torch.full((n, h, w, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    rows = L.lib.gsd_bf16_conv_partial_rows(n, h, w, cout)
    part = torch.full((rows, 2 * mp), float("nan"), dtype=torch.float32, device="cuda")
    din, dout = L.make_nhwc(xin), L.make_nhwc(out)
    L.check(L.lib.gsd_bf16_conv3x3(C.byref(din), image(L, 0, w_d, cout, cin).data_ptr(), C.byref(dout), cin, cout, part.data_ptr(),
                                   None, L.stream_ptr()), "conv")
    assert bool(torch.isfinite(out.float()).all())
    ref = F.conv2d(x[sub].double(), wq.double(), padding=1)
    assert_close_bf16(from_nhwc(out[sub], 0, cout), ref, f"{name} forward")
    sums = torch.zeros((65 * 2 * cout,), dtype=torch.float64, device="cuda")
    L.check(L.lib.gsd_bn_reduce_partials(part.data_ptr(), rows, mp, cout, sums.data_ptr(), L.stream_ptr()), "reduce")
    od = out.double()
    np.testing.assert_allclose(sums[:cout].cpu().numpy(), od.sum(dim=(0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(sums[cout:2 * cout].cpu().numpy(), (od * od).sum(dim=(0, 1, 2)).cpu().numpy(), rtol=1e-5)
    del out, od, ref

    # ---- dX of the same layer: gradient of its output -> gradient of its input
    dy = bf16r(torch.randn((n, cout, h, w), generator=g))
    dyb = to_nhwc(dy)
    ddy = L.make_nhwc(dyb)
    xs = x[sub].double().requires_grad_(True)
    F.conv2d(xs, wq.double(), padding=1).backward(dy[sub].double())
    dx_ref = xs.grad
    dz = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device="cuda")
    img_d = image(L, 1, w_d, cout, cin)
    if fused:     # second conv of a DoubleConv: dz = dX where relu(bn(y_prev)) > 0, + (sum dz, sum dz xhat) of the producer's BatchNorm
        yprev = bf16r(torch.randn((n, cin, h, w), generator=g))
        gam, bet = torch.rand((cin,), generator=g) + 0.5, 0.3 * torch.randn((cin,), generator=g)
        mean, invstd = 0.2 * torch.randn((cin,), generator=g), torch.rand((cin,), generator=g) + 0.5
        scale, shift = gam * invstd, bet - mean * gam * invstd
        dev = [t.cuda() for t in (scale, shift, mean, invstd)]
        yb = to_nhwc(yprev)
        dyv = L.make_nhwc(yb)
        mpi = L.lib.gsd_bf16_conv_mpad(cin)
        rows_d = L.lib.gsd_bf16_conv_partial_rows(n, h, w, cin)
        part_d = torch.full((rows_d, 2 * mpi), float("nan"), dtype=torch.float32, device="cuda")
        bw = L.gsd_bf16_bnbwd()
        bw.y = C.pointer(dyv)
        bw.scale, bw.shift, bw.mean, bw.invstd = (t.data_ptr() for t in dev)
        L.check(L.lib.gsd_bf16_conv3x3(C.byref(ddy), img_d.data_ptr(), C.byref(L.make_nhwc(dz)), cout, cin, part_d.data_ptr(),
                                       C.byref(bw), L.stream_ptr()), "fused dX")
        b_ = lambda v: v[None, :, None, None]      # noqa: E731
        mask = (yprev[sub] * b_(scale) + b_(shift)) > 0
        assert_close_bf16(from_nhwc(dz[sub], 0, cin), dx_ref * mask, f"{name} dX (fused)")
        got = from_nhwc(dz, 0, cin).double()
        xhat = (yprev.double() - b_(mean).double()) * b_(invstd).double()
        sd = part_d.double().sum(dim=0).cpu()
        np.testing.assert_allclose(sd[:cin].numpy(), got.sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=2e-2)
        np.testing.assert_allclose(sd[mpi:mpi + cin].numpy(), (got * xhat).sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=2e-2)
    else:
        L.check(L.lib.gsd_bf16_conv3x3(C.byref(ddy), img_d.data_ptr(), C.byref(L.make_nhwc(dz)), cout, cin, None, None,
                                       L.stream_ptr()), "dX")
        assert_close_bf16(from_nhwc(dz[sub], 0, cin), dx_ref, f"{name} dX")
    del dz, dx_ref, xs

    # ---- dW: dy against the layer's input, reduction over all N*H*W pixels
    dw_ref = torch.zeros((cout, cin, 3, 3), dtype=torch.float64)
    dt = torch.float64 if n <= 2 else torch.float32     # batch 32: fp32 BLAS sums of 4 images each, accumulated in fp64
    for i in range(0, n, 4):
        wz = torch.zeros((cout, cin, 3, 3), dtype=dt, requires_grad=True)
        F.conv2d(x[i:i + 4].to(dt), wz, padding=1).backward(dy[i:i + 4].to(dt))
        dw_ref += wz.grad.double()
    got = _wgrad(L, dyb, 0, cout, xin, 0, cin, 9, 1, T3Y, T3X, cin).reshape(cout, cin, 9).double()
    assert bool(torch.isfinite(got).all())
    assert float((got - dw_ref.reshape(cout, cin, 9)).abs().max()) <= 2e-5 * float(dw_ref.abs().max()), f"{name} dW"


CONVT = [("up0.up", 4, 1024), ("up1.up", 3, 512), ("up2.up", 2, 256), ("up3.up", 1, 128)]
CONVT_CASES = [(c, 2) for c in CONVT] + [(c, 32) for c in CONVT[:2]]


@pytest.mark.parametrize("case,n", CONVT_CASES, ids=[f"{c[0]}-N{n}" for c, n in CONVT_CASES])
def test_bf16_convT_at_network_shape(case, n):
    """ConvTranspose2d(Cin, Cin/2, 2, 2) of the four decoder levels in the bf16 engine's form: forward (+bias) scattered into
    the second half of the level's concat buffer at its F.pad offset, dX from the gradient slice, dW and the bias gradient."""
    L = _lib()
    name, lvl, cin = case
    cout, h, w = cin // 2, HS[lvl], WS[lvl]
    H2, W2 = HS[lvl - 1], WS[lvl - 1]
    oy, ox = (H2 - 2 * h) // 2, (W2 - 2 * w) // 2
    g = torch.Generator().manual_seed(cin + n)
    x = bf16r(torch.randn((n, cin, h, w), generator=g))
    wT = torch.randn((cin, cout, 2, 2), generator=g) / cin ** 0.5
    wq = bf16r(wT)
    bias = torch.randn((cout,), generator=g)
    z = L.int_array([0])
    xin = to_nhwc(x)
    cat = torch.zeros((n, H2, W2, 2 * cout), dtype=torch.bfloat16, device="cuda")
    din, dout = L.make_nhwc(xin), L.make_nhwc(cat, cout, cout)
    bias_d, w_d = bias.cuda(), wT.cuda()
    L.check(L.lib.gsd_bf16_conv_dense(C.byref(din), image(L, 3, w_d, cout, cin).data_ptr(), C.byref(dout), cin, 4 * cout, 1, 1, z, z,
                                      h, w, cout, oy, ox, bias_d.data_ptr(), None, None, L.stream_ptr()), "convT")
    ref = F.conv_transpose2d(x.double(), wq.double(), bias.double(), stride=2)
    got = cat.float().cpu()
    assert_close_bf16(got[:, oy:oy + 2 * h, ox:ox + 2 * w, cout:].permute(0, 3, 1, 2), ref, f"{name} forward")
    assert float(got[..., :cout].abs().max()) == 0.0          # the skip half is untouched
    del got, ref
    # dX: 4 taps at stride 2 over the gradient slice of the concat buffer
    gy = bf16r(torch.randn((n, cout, 2 * h, 2 * w), generator=g))
    gcat = torch.zeros((n, H2, W2, 2 * cout), dtype=torch.bfloat16)
    gcat[:, oy:oy + 2 * h, ox:ox + 2 * w, cout:] = gy.permute(0, 2, 3, 1).to(torch.bfloat16)
    gcat = gcat.cuda()
    xd = x.double().requires_grad_(True)
    wz = wq.double().clone().requires_grad_(True)
    bz = torch.zeros((cout,), dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(xd, wz, bz, stride=2).backward(gy.double())
    dx = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device="cuda")
    dgy, ddx = L.make_nhwc(gcat, cout, cout), L.make_nhwc(dx)
    ty, tx = L.int_array([oy, oy, oy + 1, oy + 1]), L.int_array([ox, ox + 1, ox, ox + 1])
    L.check(L.lib.gsd_bf16_conv_dense(C.byref(dgy), image(L, 4, w_d, cout, cin).data_ptr(), C.byref(ddx), cout, cin, 4, 2, ty, tx, h, w,
                                      0, 0, 0, None, None, None, L.stream_ptr()), "convT dX")
    assert_close_bf16(from_nhwc(dx, 0, cin), xd.grad, f"{name} dX")
    # dW (a = x, b = the gradient slice, taps (kh + oy, kw + ox) at stride 2) and the bias gradient
    gotw = _wgrad(L, xin, 0, cin, gcat, cout, cout, 4, 2, [oy, oy, oy + 1, oy + 1], [ox, ox + 1, ox, ox + 1], cout)
    gotw = gotw.reshape(cin, cout, 2, 2).double()
    assert float((gotw - wz.grad).abs().max()) <= 2e-5 * float(wz.grad.abs().max()), f"{name} dW"
    nws = L.lib.gsd_bf16_channel_sums_workspace(n, 2 * h, 2 * w, cout)
    ws, db = torch.empty((nws,), device="cuda"), torch.zeros((cout,), device="cuda")
    L.check(L.lib.gsd_bf16_channel_sums(C.byref(dgy), oy, ox, 2 * h, 2 * w, db.data_ptr(), ws.data_ptr(), nws, L.stream_ptr()), "db")
    assert float((db.cpu().double() - bz.grad).abs().max()) <= 2e-5 * float(bz.grad.abs().max()) + 1e-9, f"{name} db"


def _convT_both_kernels(L, fn):
    """fn() under GSD_BF16_CTGEMM=1 (the large-tile kernel, csrc/gsd_bf16_ctgemm.hip) and =0 (gconv_bf16_kernel<1,..>)."""
    import os
    old = os.environ.get("GSD_BF16_CTGEMM")
    out = []
    try:
        for v in ("1", "0"):
            os.environ["GSD_BF16_CTGEMM"] = v
            out.append(fn())
    finally:
        if old is None:
            os.environ.pop("GSD_BF16_CTGEMM", None)
        else:
            os.environ["GSD_BF16_CTGEMM"] = old
    return out


CT_CASES = [(CONVT[0], 2), (CONVT[1], 3), (CONVT[2], 2), (CONVT[3], 1), (CONVT[3], 2), (CONVT[0], 32), (CONVT[2], 8)]


@pytest.mark.parametrize("case,n", CT_CASES, ids=[f"{c[0]}-N{n}" for c, n in CT_CASES])
def test_bf16_convT_large_tile_kernel_equals_the_general_kernel(case, n):
    """The transposed convolutions' forward and dX on the large-tile kernel (256 x 256 / 128 x 256 tiles, eight waves) against the
    DMA-filled general kernel they ran on before: same operands, same k order through the same MFMA -> bit-identical outputs
    (scattered forward with bias; plain dX; dX with the fused BatchNorm-backward pass 1, whose sums agree to fp32 rounding)."""
    L = _lib()
    name, lvl, cin = case
    cout, h, w = cin // 2, HS[lvl], WS[lvl]
    H2, W2 = HS[lvl - 1], WS[lvl - 1]
    oy, ox = (H2 - 2 * h) // 2, (W2 - 2 * w) // 2
    g = torch.Generator().manual_seed(7 * cin + n)
    x = bf16r(torch.randn((n, cin, h, w), generator=g))
    w_d = (torch.randn((cin, cout, 2, 2), generator=g) / cin ** 0.5).cuda()
    bias_d = torch.randn((cout,), generator=g).cuda()
    xin = to_nhwc(x)
    z = L.int_array([0])
    img_f, img_d = image(L, 3, w_d, cout, cin), image(L, 4, w_d, cout, cin)

    def fwd():
        cat = torch.zeros((n, H2, W2, 2 * cout), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib.gsd_bf16_conv_dense(C.byref(L.make_nhwc(xin)), img_f.data_ptr(), C.byref(L.make_nhwc(cat, cout, cout)), cin, 4 * cout,
                                          1, 1, z, z, h, w, cout, oy, ox, bias_d.data_ptr(), None, None, L.stream_ptr()), "convT")
        torch.cuda.synchronize()
        return cat
    a, b = _convT_both_kernels(L, fwd)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"{name} forward"
    assert float(a[:, oy:oy + 2 * h, ox:ox + 2 * w, cout:].float().abs().max()) > 0
    del a, b

    gcat = torch.zeros((n, H2, W2, 2 * cout), dtype=torch.bfloat16)
    gcat[:, oy:oy + 2 * h, ox:ox + 2 * w, cout:] = bf16r(torch.randn((n, 2 * h, 2 * w, cout), generator=g)).to(torch.bfloat16)
    gcat = gcat.cuda()
    ty, tx = L.int_array([oy, oy, oy + 1, oy + 1]), L.int_array([ox, ox + 1, ox, ox + 1])
    dgy = L.make_nhwc(gcat, cout, cout)

    def dx_plain():
        dx = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib.gsd_bf16_conv_dense(C.byref(dgy), img_d.data_ptr(), C.byref(L.make_nhwc(dx)), cout, cin, 4, 2, ty, tx, h, w, 0, 0, 0,
                                          None, None, None, L.stream_ptr()), "convT dX")
        torch.cuda.synchronize()
        return dx
    a, b = _convT_both_kernels(L, dx_plain)
    assert not bool(torch.isnan(a.float()).any())
    assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"{name} dX"
    del a, b

    # fused pass 1 of the BatchNorm+ReLU backward of the unit below: dz = dX where relu(bn(y)) > 0, sums of dz and dz * xhat
    y = bf16r(torch.randn((n, h, w, cin), generator=g)).to(torch.bfloat16).cuda()
    coef = [t.cuda() for t in (torch.rand((cin,), generator=g) + 0.5, 0.3 * torch.randn((cin,), generator=g),
                               0.2 * torch.randn((cin,), generator=g), torch.rand((cin,), generator=g) + 0.5)]
    dyv = L.make_nhwc(y)
    mp = L.lib.gsd_bf16_conv_mpad(cin)

    def dx_fused():
        rows = L.lib.gsd_bf16_conv_dense_partial_rows(n, h, w, cout, cin, 4, 2)
        assert rows > 0
        dz = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device="cuda")
        part = torch.full((rows, 2 * mp), float("nan"), device="cuda")
        bw = L.gsd_bf16_bnbwd()
        bw.y = C.pointer(dyv)
        bw.scale, bw.shift, bw.mean, bw.invstd = (t.data_ptr() for t in coef)
        L.check(L.lib.gsd_bf16_conv_dense(C.byref(dgy), img_d.data_ptr(), C.byref(L.make_nhwc(dz)), cout, cin, 4, 2, ty, tx, h, w, 0, 0, 0,
                                          None, part.data_ptr(), C.byref(bw), L.stream_ptr()), "convT dX fused")
        torch.cuda.synchronize()
        return dz, part.double().sum(dim=0).cpu()
    (za, sa), (zb, sb) = _convT_both_kernels(L, dx_fused)
    assert torch.equal(za.view(torch.int16), zb.view(torch.int16)), f"{name} fused dz"
    assert bool(torch.isfinite(sa[:cin]).all()) and bool(torch.isfinite(sa[mp:mp + cin]).all())
    # reference sums in fp64 from the stored dz
    zf, yf = za.double().cpu(), y.double().cpu()
    xhat = (yf - coef[2].double().cpu()) * coef[3].double().cpu()
    r1, r2 = zf.sum(dim=(0, 1, 2)), (zf * xhat).sum(dim=(0, 1, 2))
    scale1, scale2 = float(zf.abs().sum(dim=(0, 1, 2)).max()), float((zf * xhat).abs().sum(dim=(0, 1, 2)).max())
    for s_ in (sa, sb):
        assert float((s_[:cin] - r1).abs().max()) <= 2e-5 * scale1, f"{name} sum dz"
        assert float((s_[mp:mp + cin] - r2).abs().max()) <= 2e-5 * scale2, f"{name} sum dz*xhat"
