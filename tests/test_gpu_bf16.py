"""GPU parity of the bf16 mixed-precision kernel family (include/gsd_bf16.h) against fp64 torch-CPU evaluations of the
same contractions on the SAME bf16-rounded operands: what is left is fp32 accumulation order and the final rounding of
the result to bf16 (half an ulp = 2^-9 relative)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16_TOL = 2.0 ** -8      # one bf16 ulp, relative, per element (rounding of the stored result + accumulation noise)


def _lib():
    from gelslim_depth_amd import _lib as L
    return L


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def to_nhwc(x, c_total=None, c_off=0):
    """(N,C,H,W) float -> contiguous (N,H,W,c_total) bf16 CUDA buffer holding x in channels [c_off, c_off+C)."""
    n, c, h, w = x.shape
    ct = c if c_total is None else c_total
    buf = torch.full((n, h, w, ct), 7.0, dtype=torch.bfloat16)       # 7: a wrong read of a neighbouring slice shows up
    buf[..., c_off:c_off + c] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    return buf.cuda()


def from_nhwc(buf, c_off, c):
    return buf[..., c_off:c_off + c].float().cpu().permute(0, 3, 1, 2).contiguous()


def weight_image(w_tmk):
    """(T, M, K) float -> (T, mpad(M), K) bf16 on the GPU, zero padded rows."""
    L = _lib()
    t, m, k = w_tmk.shape
    mp = L.lib.gsd_bf16_conv_mpad(m)
    img = torch.zeros((t, mp, k), dtype=torch.bfloat16)
    img[:, :m] = w_tmk.to(torch.bfloat16)
    return img.cuda()


def assert_close_bf16(got, ref, what):
    err = (got.double() - ref.double()).abs()
    tol = BF16_TOL * ref.double().abs() + 1e-3 * float(ref.abs().max()) * BF16_TOL * 4
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} off, worst {float((err / (ref.abs() + 1e-6)).max()):.3e}"


@pytest.mark.parametrize("n,h,w,k,m,in_tot,in_off,out_tot,out_off", [
    (2, 20, 26, 64, 128, 64, 0, 128, 0),        # 128x256 tile, bottom-level extent, K two chunks
    (1, 33, 70, 32, 64, 96, 32, 64, 0),         # 64x512 tile, input is a channel slice
    (2, 17, 21, 96, 48, 96, 0, 80, 16),         # M not a multiple of 64, output is a channel slice, 3 chunks
    (1, 40, 53, 128, 256, 128, 0, 256, 0),      # two m-blocks
    (2, 64, 64, 32, 256, 32, 0, 256, 0),        # 64 work items, two m-blocks: the XCD-aware item order is on (grid % 8 == 0)
    (8, 64, 128, 32, 256, 32, 0, 256, 0),       # 512 items on 256 persistent blocks: two items per block in XCD order
    (2, 32, 64, 64, 512, 64, 0, 512, 0),        # four m-blocks in XCD order
])
def test_conv3x3_bf16(n, h, w, k, m, in_tot, in_off, out_tot, out_off):
    L = _lib()
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + m)
    x = bf16r(torch.randn((n, k, h, w), generator=g))
    wt = bf16r(torch.randn((m, k, 3, 3), generator=g) / (3.0 * k ** 0.5))
    ref = F.conv2d(x.double(), wt.double(), padding=1)
    xin = to_nhwc(x, in_tot, in_off)
    out = torch.full((n, h, w, out_tot), 5.0, dtype=torch.bfloat16, device="cuda")
    img = weight_image(wt.permute(2, 3, 0, 1).reshape(9, m, k))
    rows = L.lib.gsd_bf16_conv_partial_rows(n, h, w, m)
    mp = L.lib.gsd_bf16_conv_mpad(m)
    part = torch.zeros((rows, 2 * mp), dtype=torch.float32, device="cuda")
    din, dout = L.make_nhwc(xin, in_off, k), L.make_nhwc(out, out_off, m)
    L.check(L.lib.gsd_bf16_conv3x3(C.byref(din), img.data_ptr(), C.byref(dout), k, m, part.data_ptr(), None, L.stream_ptr()), "conv")
    torch.cuda.synchronize()
    got = from_nhwc(out, out_off, m)
    assert_close_bf16(got, ref, "conv3x3")
    # channels outside the slice are untouched
    if out_tot > m:
        rest = torch.cat([out[..., :out_off], out[..., out_off + m:]], dim=-1).float()
        assert bool((rest == 5.0).all())
    # BatchNorm partial sums are those of the stored values
    sums = torch.zeros((65 * 2 * m,), dtype=torch.float64, device="cuda")
    L.check(L.lib.gsd_bn_reduce_partials(part.data_ptr(), rows, mp, m, sums.data_ptr(), L.stream_ptr()), "reduce")
    s = sums[:2 * m].cpu()
    assert torch.allclose(s[:m], got.double().sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[m:], (got.double() ** 2).sum(dim=(0, 2, 3)), rtol=1e-5, atol=1e-3)


def test_conv3x3_bf16_item_order_is_only_an_order(monkeypatch):
    """GSD_BF16_XCD (XCD-aware order of the persistent blocks' work items) changes which block computes which pixel tile,
    nothing else: outputs bit-identical, BatchNorm sums equal to rounding, for the plain and the fused-BatchNorm-backward launch."""
    L = _lib()
    g = torch.Generator().manual_seed(77)
    n, h, w, k, m = 4, 64, 128, 64, 256
    x = to_nhwc(bf16r(torch.randn((n, k, h, w), generator=g)))
    img = weight_image(bf16r(torch.randn((9, m, k), generator=g) / (3.0 * k ** 0.5)))
    ybw = to_nhwc(bf16r(torch.randn((n, m, h, w), generator=g)))
    vec = [(torch.rand(m, generator=g) + 0.5).cuda(), (torch.randn(m, generator=g) * 0.3).cuda(),
           (torch.randn(m, generator=g) * 0.3).cuda(), (torch.rand(m, generator=g) + 0.5).cuda()]
    rows, mp = L.lib.gsd_bf16_conv_partial_rows(n, h, w, m), L.lib.gsd_bf16_conv_mpad(m)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("GSD_BF16_XCD", flag)
        outs = []
        for fused in (False, True):
            out = torch.full((n, h, w, m), float("nan"), dtype=torch.bfloat16, device="cuda")
            part = torch.full((rows, 2 * mp), float("nan"), dtype=torch.float32, device="cuda")
            din, dout = L.make_nhwc(x), L.make_nhwc(out)
            bw = None
            if fused:
                bw = L.gsd_bf16_bnbwd()
                dyb = L.make_nhwc(ybw)
                bw.y = C.pointer(dyb)
                bw.scale, bw.shift, bw.mean, bw.invstd = [v.data_ptr() for v in vec]
                bw = C.byref(bw)
            L.check(L.lib.gsd_bf16_conv3x3(C.byref(din), img.data_ptr(), C.byref(dout), k, m, part.data_ptr(), bw, L.stream_ptr()), "conv")
            sums = torch.zeros((65 * 2 * m,), dtype=torch.float64, device="cuda")
            L.check(L.lib.gsd_bn_reduce_partials(part.data_ptr(), rows, mp, m, sums.data_ptr(), L.stream_ptr()), "reduce")
            outs += [out, sums[:2 * m].clone()]
        res[flag] = outs
    for i in (0, 2):
        assert bool(torch.isfinite(res["1"][i].float()).all())
        assert torch.equal(res["0"][i].view(torch.int16), res["1"][i].view(torch.int16))
    for i in (1, 3):
        assert bool(torch.isfinite(res["1"][i]).all())
        np.testing.assert_allclose(res["1"][i].cpu().numpy(), res["0"][i].cpu().numpy(), rtol=1e-5, atol=1e-2)


def test_weight_images_batched_equal_one_by_one():
    """gsd_bf16_weight_images (every image of a step in one launch per 32 jobs) writes what gsd_bf16_weight_image writes job by
    job -- all five layouts, 40 jobs (two launches), sizes with and without padding."""
    L = _lib()
    g = torch.Generator().manual_seed(99)
    shapes = [(0, 64, 3), (0, 64, 64), (1, 64, 64), (2, 64, 3), (3, 32, 64), (4, 32, 64), (0, 128, 96), (1, 128, 96), (0, 48, 160),
              (3, 64, 128), (3, 24, 64), (4, 24, 40)] * 3 + [(0, 64, 3), (1, 512, 256), (3, 256, 512), (4, 256, 512)]
    jobs = (L.gsd_bf16_wimg_job * len(shapes))()
    keep, one_by_one = [], []
    for i, (mode, cout, cin) in enumerate(shapes):
        kk = 2 if mode >= 3 else 3
        w = (torch.randn((cin, cout, kk, kk) if mode >= 3 else (cout, cin, kk, kk), generator=g)).cuda()
        n = L.lib.gsd_bf16_weight_image_size(mode, cout, cin)
        a = torch.full((n,), float("nan"), dtype=torch.bfloat16, device="cuda")
        b = torch.full((n,), float("nan"), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib.gsd_bf16_weight_image(mode, w.data_ptr(), cout, cin, a.data_ptr(), L.stream_ptr()), "weight_image")
        jobs[i].w, jobs[i].out, jobs[i].mode, jobs[i].Cout, jobs[i].Cin = w.data_ptr(), b.data_ptr(), mode, cout, cin
        keep.append(w)
        one_by_one.append((a, b))
    L.check(L.lib.gsd_bf16_weight_images(jobs, len(shapes), L.stream_ptr()), "weight_images")
    for a, b in one_by_one:
        assert bool(torch.isfinite(b.float()).all())
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    assert L.lib.gsd_bf16_weight_images(jobs, 0, L.stream_ptr()) != 0      # refused: no jobs


def test_dense_1x1_and_convT_bf16():
    L = _lib()
    g = torch.Generator().manual_seed(5)
    # 1x1 (the im2col'd first layer): K 32 -> M 64
    n, h, w, k, m = 2, 21, 37, 32, 64
    x = bf16r(torch.randn((n, k, h, w), generator=g))
    wt = bf16r(torch.randn((m, k), generator=g) / k ** 0.5)
    ref = torch.einsum("nkhw,mk->nmhw", x.double(), wt.double())
    xin, out = to_nhwc(x), torch.zeros((n, h, w, m), dtype=torch.bfloat16, device="cuda")
    din, dout = L.make_nhwc(xin), L.make_nhwc(out)
    z = L.int_array([0])
    img1 = weight_image(wt[None])
    L.check(L.lib.gsd_bf16_conv_dense(C.byref(din), img1.data_ptr(), C.byref(dout), k, m, 1, 1, z, z, h, w, 0, 0,
                                      0, None, None, None, L.stream_ptr()), "1x1")
    assert_close_bf16(from_nhwc(out, 0, m), ref, "1x1")

    # ConvTranspose2d(k2,s2)+bias scattered into the second half of a concat buffer, with the F.pad offset (0,1)
    n, h, w, cin, cout = 2, 10, 13, 64, 32
    x = bf16r(torch.randn((n, cin, h, w), generator=g))
    wT = bf16r(torch.randn((cin, cout, 2, 2), generator=g) / cin ** 0.5)
    bias = torch.randn((cout,), generator=g)
    ref = F.conv_transpose2d(x.double(), wT.double(), bias.double(), stride=2)            # (n, cout, 2h, 2w)
    cat = torch.zeros((n, 2 * h + 1, 2 * w + 1, 2 * cout), dtype=torch.bfloat16, device="cuda")
    img = weight_image(wT.permute(2, 3, 1, 0).reshape(1, 4 * cout, cin))                    # m = (kh*2+kw)*cout + co
    xin = to_nhwc(x)
    din, dout = L.make_nhwc(xin), L.make_nhwc(cat, cout, cout)
    bias_d = bias.cuda()
    L.check(L.lib.gsd_bf16_conv_dense(C.byref(din), img.data_ptr(), C.byref(dout), cin, 4 * cout, 1, 1, z, z, h, w, cout, 0, 1,
                                      bias_d.data_ptr(), None, None, L.stream_ptr()), "convT")
    got = cat.float().cpu()
    assert_close_bf16(got[:, 0:2 * h, 1:2 * w + 1, cout:].permute(0, 3, 1, 2), ref, "convT")
    assert float(got[..., :cout].abs().max()) == 0.0 and float(got[:, 2 * h:].abs().max()) == 0.0
    assert float(got[:, :, 0].abs().max()) == 0.0

    # its dX: 4 taps at stride 2 over the (cropped) gradient slice
    gy = bf16r(torch.randn((n, cout, 2 * h, 2 * w), generator=g))
    gcat = torch.zeros((n, 2 * h + 1, 2 * w + 1, 2 * cout), dtype=torch.bfloat16)
    gcat[:, 0:2 * h, 1:2 * w + 1, cout:] = gy.permute(0, 2, 3, 1).to(torch.bfloat16)
    gcat = gcat.cuda()
    xd = x.double().requires_grad_(True)
    F.conv_transpose2d(xd, wT.double(), None, stride=2).backward(gy.double())
    img_d = weight_image(wT.permute(2, 3, 0, 1).reshape(4, cin, cout))                      # [q][ci][co]
    dx = torch.zeros((n, h, w, cin), dtype=torch.bfloat16, device="cuda")
    din, dout = L.make_nhwc(gcat, cout, cout), L.make_nhwc(dx)
    ty, tx = L.int_array([0, 0, 1, 1]), L.int_array([1, 2, 1, 2])                            # (kh, kw + pad offset 1)
    L.check(L.lib.gsd_bf16_conv_dense(C.byref(din), img_d.data_ptr(), C.byref(dout), cout, cin, 4, 2, ty, tx, h, w, 0, 0, 0,
                                      None, None, None, L.stream_ptr()), "convT dgrad")
    assert_close_bf16(from_nhwc(dx, 0, cin), xd.grad, "convT dX")


def _wgrad(L, a_buf, a_off, m, b_buf, b_off, ncols, ntaps, stride, ty, tx, ncols_out):
    da, db = L.make_nhwc(a_buf, a_off, m), L.make_nhwc(b_buf, b_off, ncols)
    n, h, w = a_buf.shape[0], a_buf.shape[1], a_buf.shape[2]
    ws_n = L.lib.gsd_bf16_wgrad_workspace(ntaps, n, h, w, m, ncols)
    ws = torch.empty((ws_n,), dtype=torch.float32, device="cuda")
    dw = torch.full((m * ncols_out * ntaps,), 3.0, dtype=torch.float32, device="cuda")
    L.check(L.lib.gsd_bf16_wgrad(C.byref(da), C.byref(db), ntaps, stride, L.int_array(ty), L.int_array(tx), dw.data_ptr(),
                                 ncols_out, ws.data_ptr(), ws_n, L.stream_ptr()), "wgrad")
    return dw.cpu()


T3Y = [t // 3 - 1 for t in range(9)]
T3X = [t % 3 - 1 for t in range(9)]


@pytest.mark.parametrize("n,h,w,cin,cout,b_tot,b_off", [
    (2, 20, 45, 64, 128, 64, 0),       # 128 x 32 tiles, two n-blocks
    (1, 9, 70, 96, 64, 96, 0),         # 64 x 64 tiles (M <= 64), ragged n-block
    (2, 21, 27, 32, 48, 96, 64),       # M not a multiple of the tile, B a channel slice of a concat buffer
    (1, 40, 53, 128, 256, 128, 0),     # two m-blocks
])
def test_wgrad_conv3x3_bf16(n, h, w, cin, cout, b_tot, b_off):
    L = _lib()
    g = torch.Generator().manual_seed(h * 100 + cout)
    a_in = bf16r(torch.randn((n, cin, h, w), generator=g))
    dy = bf16r(torch.randn((n, cout, h, w), generator=g))
    wt = torch.zeros((cout, cin, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv2d(a_in.double(), wt, padding=1).backward(dy.double())
    got = _wgrad(L, to_nhwc(dy), 0, cout, to_nhwc(a_in, b_tot, b_off), b_off, cin, 9, 1, T3Y, T3X, cin).reshape(cout, cin, 3, 3)
    ref = wt.grad
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), "conv3x3 dW"


def test_wgrad_first_layer_and_convT_bf16():
    L = _lib()
    g = torch.Generator().manual_seed(9)
    # first layer: dy (64 ch) against the im2col'd 3-channel input produced by gsd_bf16_im2col3x3
    n, h, w = 2, 19, 33
    x = torch.rand((n, 3, h, w), generator=g)
    dy = bf16r(torch.randn((n, 64, h, w), generator=g))
    col = torch.zeros((n, h, w, 32), dtype=torch.bfloat16, device="cuda")
    dcol = L.make_nhwc(col)
    x_d = x.cuda()
    L.check(L.lib.gsd_bf16_im2col3x3(x_d.data_ptr(), n, 3, h, w, C.byref(dcol), L.stream_ptr()), "im2col")
    unf = F.unfold(x, 3, padding=1).reshape(n, 27, h, w)                       # k = c*9 + t
    colr = col.float().cpu()
    assert torch.equal(colr[..., :27], bf16r(unf).permute(0, 2, 3, 1)) and float(colr[..., 27:].abs().max()) == 0.0
    wt = torch.zeros((64, 3, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv2d(bf16r(x).double(), wt, padding=1).backward(dy.double())
    got = _wgrad(L, to_nhwc(dy), 0, 64, col, 0, 32, 1, 1, [0], [0], 27).reshape(64, 3, 3, 3)
    assert float((got.double() - wt.grad).abs().max()) <= 2e-5 * float(wt.grad.abs().max())
    # ConvTranspose2d dW: a = x (64 ch, low res), b = gradient slice of the concat buffer at pad offset (1, 0)
    n, h, w, cin, cout = 2, 10, 13, 64, 32
    xs = bf16r(torch.randn((n, cin, h, w), generator=g))
    gy = bf16r(torch.randn((n, cout, 2 * h, 2 * w), generator=g))
    gcat = torch.zeros((n, 2 * h + 1, 2 * w, 2 * cout), dtype=torch.bfloat16)
    gcat[:, 1:2 * h + 1, :, cout:] = gy.permute(0, 2, 3, 1).to(torch.bfloat16)
    wT = torch.zeros((cin, cout, 2, 2), dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(xs.double(), wT, None, stride=2).backward(gy.double())
    got = _wgrad(L, to_nhwc(xs), 0, cin, gcat.cuda(), cout, cout, 4, 2, [1, 1, 2, 2], [0, 1, 0, 1], cout).reshape(cin, cout, 2, 2)
    assert float((got.double() - wT.grad).abs().max()) <= 2e-5 * float(wT.grad.abs().max())


def test_weight_images_bf16():
    L = _lib()
    g = torch.Generator().manual_seed(3)
    cout, cin = 48, 40
    w = torch.randn((cout, cin, 3, 3), generator=g)
    wT = torch.randn((cin, cout, 2, 2), generator=g)
    w0 = torch.randn((cout, 3, 3, 3), generator=g)

    def image(mode, src, co, ci):
        nel = L.lib.gsd_bf16_weight_image_size(mode, co, ci)
        out = torch.full((nel,), 9.0, dtype=torch.bfloat16, device="cuda")
        src_d = src.cuda()
        L.check(L.lib.gsd_bf16_weight_image(mode, src_d.data_ptr(), co, ci, out.data_ptr(), L.stream_ptr()), "wimg")
        torch.cuda.synchronize()
        return out.float().cpu()
    r32 = lambda v: (v + 31) // 32 * 32          # noqa: E731
    r128 = lambda v: (v + 127) // 128 * 128      # noqa: E731
    i0 = image(0, w, cout, cin).reshape(9, r128(cout), r32(cin))
    assert torch.equal(i0[:, :cout, :cin], bf16r(w).permute(2, 3, 0, 1).reshape(9, cout, cin))
    assert float(i0[:, cout:].abs().max()) == 0.0 and float(i0[:, :, cin:].abs().max()) == 0.0
    i1 = image(1, w, cout, cin).reshape(9, r128(cin), r32(cout))
    assert torch.equal(i1[:, :cin, :cout], bf16r(w).flip(2, 3).permute(2, 3, 1, 0).reshape(9, cin, cout))
    i2 = image(2, w0, cout, 3).reshape(1, r128(cout), 32)
    assert torch.equal(i2[0, :cout, :27], bf16r(w0).reshape(cout, 27)) and float(i2[0, :, 27:].abs().max()) == 0.0
    i3 = image(3, wT, cout, cin).reshape(1, r128(4 * cout), r32(cin))
    assert torch.equal(i3[0, :4 * cout, :cin], bf16r(wT).permute(2, 3, 1, 0).reshape(4 * cout, cin))
    i4 = image(4, wT, cout, cin).reshape(4, r128(cin), r32(cout))
    assert torch.equal(i4[:, :cin, :cout], bf16r(wT).permute(2, 3, 0, 1).reshape(4, cin, cout))


def test_pointwise_forward_bf16():
    L = _lib()
    g = torch.Generator().manual_seed(4)
    n, h, w, c = 2, 11, 15, 40
    y = bf16r(torch.randn((n, c, h, w), generator=g))
    scale, shift = torch.rand((c,), generator=g) + 0.5, torch.randn((c,), generator=g)
    ybuf = to_nhwc(y)
    cat = torch.zeros((n, h, w, c + 24), dtype=torch.bfloat16, device="cuda")
    dy_, da_ = L.make_nhwc(ybuf), L.make_nhwc(cat, 0, c)
    sc_d, sh_d = scale.cuda(), shift.cuda()          # keep device copies alive across the launch
    L.check(L.lib.gsd_bf16_bn_apply(C.byref(dy_), sc_d.data_ptr(), sh_d.data_ptr(), C.byref(da_), 1, L.stream_ptr()), "bn_apply")
    a_ref = bf16r(torch.relu(torch.addcmul(shift[None, :, None, None], y, scale[None, :, None, None])))
    a_got = from_nhwc(cat, 0, c)
    assert float((a_got - a_ref).abs().max()) <= 2.0 ** -7 * float(a_ref.abs().max())     # fma vs mul+add: <= 1 ulp
    assert float(cat[..., c:].float().abs().max()) == 0.0
    pooled = torch.zeros((n, h // 2, w // 2, c), dtype=torch.bfloat16, device="cuda")
    dp_ = L.make_nhwc(pooled)
    L.check(L.lib.gsd_bf16_maxpool2(C.byref(da_), C.byref(dp_), L.stream_ptr()), "maxpool")
    assert torch.equal(from_nhwc(pooled, 0, c), F.max_pool2d(a_got, 2))
    wout, bout = torch.randn((1, c), generator=g), torch.randn((1,), generator=g)
    out = torch.zeros((n, 1, h, w), device="cuda")
    wo_d, bo_d = wout.cuda(), bout.cuda()
    L.check(L.lib.gsd_bf16_conv1x1_out(C.byref(da_), wo_d.data_ptr(), bo_d.data_ptr(), 1, out.data_ptr(), L.stream_ptr()), "outc")
    ref = F.conv2d(a_got.double(), wout.double()[:, :, None, None], bout.double())
    assert float((out.cpu().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("h,w", [(11, 15), (12, 16), (2, 3), (7, 8), (20, 27)])
def test_bn_apply_pool_is_apply_then_pool_bf16(h, w):
    """gsd_bf16_bn_apply_pool (BatchNorm apply + ReLU + MaxPool2d(2) from one read of y) == gsd_bf16_bn_apply followed by
    gsd_bf16_maxpool2, bit for bit, on odd and even sizes (floor mode drops the odd last row / column), writing the activation
    into a channel slice of a wider (concat) buffer and touching nothing else of it."""
    L = _lib()
    g = torch.Generator().manual_seed(40 + h)
    n, c = 3, 40
    y = bf16r(torch.randn((n, c, h, w), generator=g))
    scale, shift = torch.rand((c,), generator=g) + 0.5, torch.randn((c,), generator=g)
    ybuf = to_nhwc(y)
    sc_d, sh_d = scale.cuda(), shift.cuda()
    res = []
    for fused in (False, True):
        cat = torch.full((n, h, w, c + 24), 7.0, dtype=torch.bfloat16, device="cuda")
        pooled = torch.full((n, h // 2, w // 2, c), -3.0, dtype=torch.bfloat16, device="cuda")
        dy_, da_, dp_ = L.make_nhwc(ybuf), L.make_nhwc(cat, 0, c), L.make_nhwc(pooled)
        if fused:
            L.check(L.lib.gsd_bf16_bn_apply_pool(C.byref(dy_), sc_d.data_ptr(), sh_d.data_ptr(), C.byref(da_), C.byref(dp_),
                                                 L.stream_ptr()), "bn_apply_pool")
        else:
            L.check(L.lib.gsd_bf16_bn_apply(C.byref(dy_), sc_d.data_ptr(), sh_d.data_ptr(), C.byref(da_), 1, L.stream_ptr()), "bn_apply")
            L.check(L.lib.gsd_bf16_maxpool2(C.byref(da_), C.byref(dp_), L.stream_ptr()), "maxpool")
        torch.cuda.synchronize()
        res.append((cat.clone(), pooled.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert float((res[1][0][..., c:].float() - 7.0).abs().max()) == 0.0
    assert torch.equal(from_nhwc(res[1][1], 0, c), F.max_pool2d(from_nhwc(res[1][0], 0, c), 2))


@pytest.mark.parametrize("h,w", [(13, 18), (12, 17), (9, 11), (8, 10), (2, 2)])
def test_pool_argmax_index_routes_like_the_stored_activations_bf16(h, w):
    """gsd_bf16_bn_apply_pool_idx leaves, beside the activation and its max-pool, two bits per pooled element saying which of the
    window's four stored activations it is (first maximum wins; ties made on purpose here: a quarter of the raw values are equal
    and many activations are 0).  gsd_bf16_bn_bwd_reduce_pool_idx routing the pooled gradient by those codes == mode 1 of
    gsd_bf16_bn_bwd_reduce re-reading the activations: same dz, same partial sums, bit for bit; and the codes are the arg-max."""
    L = _lib()
    g = torch.Generator().manual_seed(70 + h)
    n, c = 3, 48
    y = bf16r(torch.round(torch.randn((n, c, h, w), generator=g) * 4) / 4)          # coarse values: ties inside windows
    gamma, beta = torch.rand((c,), generator=g) + 0.5, 0.3 * torch.randn((c,), generator=g)
    mean, var = y.mean(dim=(0, 2, 3)), y.var(dim=(0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    dev = [t.float().cuda() for t in (scale, shift, mean, invstd)]
    ybuf = to_nhwc(y)
    a = torch.zeros((n, h, w, c), dtype=torch.bfloat16, device="cuda")
    a2 = torch.zeros_like(a)
    pooled = torch.zeros((n, h // 2, w // 2, c), dtype=torch.bfloat16, device="cuda")
    pooled2 = torch.zeros_like(pooled)
    idx = torch.full((n, h // 2, w // 2, c // 8), -1, dtype=torch.int16, device="cuda")
    dy_ = L.make_nhwc(ybuf)
    L.check(L.lib.gsd_bf16_bn_apply_pool_idx(C.byref(dy_), dev[0].data_ptr(), dev[1].data_ptr(), C.byref(L.make_nhwc(a)),
                                             C.byref(L.make_nhwc(pooled)), idx.data_ptr(), L.stream_ptr()), "apply_pool_idx")
    L.check(L.lib.gsd_bf16_bn_apply_pool(C.byref(dy_), dev[0].data_ptr(), dev[1].data_ptr(), C.byref(L.make_nhwc(a2)),
                                         C.byref(L.make_nhwc(pooled2)), L.stream_ptr()), "apply_pool")
    torch.cuda.synchronize()
    assert torch.equal(a, a2) and torch.equal(pooled, pooled2)
    # the codes against a host arg-max over the stored activations (first maximum in (0,0),(0,1),(1,0),(1,1) order)
    af = a.float().cpu()[:, :h // 2 * 2, :w // 2 * 2]
    win = torch.stack([af[:, 0::2, 0::2], af[:, 0::2, 1::2], af[:, 1::2, 0::2], af[:, 1::2, 1::2]], dim=0)     # (4, n, hp, wp, c)
    ref_bi = torch.zeros(win.shape[1:], dtype=torch.int64)
    best = win[0].clone()
    for q in range(1, 4):
        better = win[q] > best
        ref_bi[better] = q
        best = torch.where(better, win[q], best)
    codes = idx.cpu().to(torch.int32) & 0xffff
    got_bi = torch.stack([(codes >> (2 * i)) & 3 for i in range(8)], dim=-1).reshape(n, h // 2, w // 2, c).to(torch.int64)
    assert torch.equal(got_bi, ref_bi)
    # backward: the two routings
    gsk = bf16r(torch.randn((n, h, w, c), generator=g)).to(torch.bfloat16).cuda()
    dpool = bf16r(torch.randn((n, h // 2, w // 2, c), generator=g)).to(torch.bfloat16).cuda()
    rows = L.lib.gsd_bf16_bn_bwd_partial_rows(n, h, w)
    res = []
    for use_idx in (False, True):
        dz = torch.full((n, h, w, c), float("nan"), dtype=torch.bfloat16, device="cuda")
        part = torch.full((rows, 3 * c), float("nan"), device="cuda")
        args = (C.byref(dy_), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), C.byref(L.make_nhwc(gsk)))
        if use_idx:
            L.check(L.lib.gsd_bf16_bn_bwd_reduce_pool_idx(*args, idx.data_ptr(), C.byref(L.make_nhwc(dpool)), C.byref(L.make_nhwc(dz)),
                                                          part.data_ptr(), L.stream_ptr()), "reduce_pool_idx")
        else:
            L.check(L.lib.gsd_bf16_bn_bwd_reduce(1, *args, C.byref(L.make_nhwc(a)), C.byref(L.make_nhwc(dpool)), None, None,
                                                 C.byref(L.make_nhwc(dz)), part.data_ptr(), L.stream_ptr()), "reduce_pool")
        torch.cuda.synchronize()
        res.append((dz.clone(), part[:, :2 * c].clone()))
    assert not bool(torch.isnan(res[1][0].float()).any())
    assert torch.equal(res[0][0].view(torch.int16), res[1][0].view(torch.int16))
    assert torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("mode,h,w", [(0, 13, 18), (1, 13, 18), (1, 12, 17), (1, 9, 11), (1, 8, 10), (2, 13, 18)])
def test_bn_bwd_bf16(mode, h, w):
    """Pass 1 (mask, pooled-gradient routing, output-conv gradient, per-channel sums) and pass 2 against torch fp64."""
    L = _lib()
    g = torch.Generator().manual_seed(10 + mode + h)
    n, c = 2, 72
    y = bf16r(torch.randn((n, c, h, w), generator=g))
    gamma, beta = torch.rand((c,), generator=g) + 0.5, 0.3 * torch.randn((c,), generator=g)
    mean = y.mean(dim=(0, 2, 3))
    var = y.var(dim=(0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    yn = torch.addcmul(shift[None, :, None, None], y, scale[None, :, None, None])
    a = bf16r(torch.relu(yn))
    dev = lambda t: t.cuda()     # noqa: E731
    ybuf, abuf = to_nhwc(y), to_nhwc(a)
    gsrc = bf16r(torch.randn((n, c, h, w), generator=g))
    dpool = bf16r(torch.randn((n, c, h // 2, w // 2), generator=g))
    dout = torch.randn((n, 1, h, w), generator=g)
    wout = torch.randn((1, c), generator=g)
    if mode == 0:
        da = gsrc.clone()
    elif mode == 1:
        ad = a.double().requires_grad_(True)
        F.max_pool2d(ad, 2).backward(dpool.double())          # torch routes to the first maximum, like the kernel
        da = gsrc + ad.grad.float()
    else:
        da = dout * wout[0][None, :, None, None]
    dz_ref = torch.where(yn > 0, da, torch.zeros_like(da))
    gbuf, pbuf = to_nhwc(gsrc), to_nhwc(dpool)
    dzbuf = torch.zeros((n, h, w, c), dtype=torch.bfloat16, device="cuda")
    rows = L.lib.gsd_bf16_bn_bwd_partial_rows(n, h, w)
    part = torch.zeros((rows, 3 * c), dtype=torch.float32, device="cuda")
    keep = [dev(scale), dev(shift), dev(mean), dev(invstd), dev(dout), dev(wout)]
    L.check(L.lib.gsd_bf16_bn_bwd_reduce(mode, C.byref(L.make_nhwc(ybuf)), keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(),
                                         keep[3].data_ptr(), C.byref(L.make_nhwc(gbuf)), C.byref(L.make_nhwc(abuf)),
                                         C.byref(L.make_nhwc(pbuf)), keep[4].data_ptr(), keep[5].data_ptr(),
                                         C.byref(L.make_nhwc(dzbuf)), part.data_ptr(), L.stream_ptr()), "bn_bwd_reduce")
    dz_got = from_nhwc(dzbuf, 0, c)
    assert float((dz_got - bf16r(dz_ref)).abs().max()) <= 2.0 ** -7 * float(dz_ref.abs().max())
    sums = part.double().sum(dim=0).cpu()
    xhat = (y - mean[None, :, None, None]) * invstd[None, :, None, None]
    assert torch.allclose(sums[:c], dz_got.double().sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-3)
    assert torch.allclose(sums[c:2 * c], (dz_got.double() * xhat.double()).sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-3)
    if mode == 2:
        assert torch.allclose(sums[2 * c:], (dout.double() * a.double()).sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-3)
    cnt = float(n * h * w)
    c1, c2 = (sums[:c] / cnt).float(), (sums[c:2 * c] / cnt).float()
    keep2 = [dev(c1), dev(c2)]
    L.check(L.lib.gsd_bf16_bn_bwd_apply(C.byref(L.make_nhwc(dzbuf)), C.byref(L.make_nhwc(ybuf)), keep[0].data_ptr(), keep[2].data_ptr(),
                                        keep[3].data_ptr(), keep2[0].data_ptr(), keep2[1].data_ptr(), L.stream_ptr()), "bn_bwd_apply")
    ref = scale[None, :, None, None] * (dz_got - c1[None, :, None, None] - xhat * c2[None, :, None, None])
    got = from_nhwc(dzbuf, 0, c)
    assert float((got - bf16r(ref)).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())


def test_conv3x3_dgrad_through_weight_image_bf16():
    """dX of conv3x3 == conv3x3 of dy with the mode-1 weight image (taps flipped, in/out channels swapped)."""
    L = _lib()
    g = torch.Generator().manual_seed(12)
    n, h, w, cin, cout = 2, 14, 19, 64, 96
    wt = torch.randn((cout, cin, 3, 3), generator=g) / (3.0 * cin ** 0.5)
    dy = bf16r(torch.randn((n, cout, h, w), generator=g))
    xz = torch.zeros((n, cin, h, w), dtype=torch.float64, requires_grad=True)
    F.conv2d(xz, bf16r(wt).double(), padding=1).backward(dy.double())
    img = torch.empty((L.lib.gsd_bf16_weight_image_size(1, cout, cin),), dtype=torch.bfloat16, device="cuda")
    wd = wt.cuda()
    L.check(L.lib.gsd_bf16_weight_image(1, wd.data_ptr(), cout, cin, img.data_ptr(), L.stream_ptr()), "wimg")
    dyb, dx = to_nhwc(dy), torch.zeros((n, h, w, cin), dtype=torch.bfloat16, device="cuda")
    din, dout = L.make_nhwc(dyb), L.make_nhwc(dx)
    L.check(L.lib.gsd_bf16_conv3x3(C.byref(din), img.data_ptr(), C.byref(dout), cout, cin, None, None, L.stream_ptr()), "dgrad")
    assert_close_bf16(from_nhwc(dx, 0, cin), xz.grad, "conv3x3 dX")


def test_channel_sums_bf16():
    L = _lib()
    g = torch.Generator().manual_seed(13)
    n, h, w, c = 3, 21, 27, 32
    t = bf16r(torch.randn((n, c, h, w), generator=g))
    buf = to_nhwc(t, 64, 32)
    view = L.make_nhwc(buf, 32, c)
    nws = L.lib.gsd_bf16_channel_sums_workspace(n, 20, 26, c)
    ws, out = torch.empty((nws,), device="cuda"), torch.zeros((c,), device="cuda")
    L.check(L.lib.gsd_bf16_channel_sums(C.byref(view), 1, 0, 20, 26, out.data_ptr(), ws.data_ptr(), nws, L.stream_ptr()), "sums")
    ref = t[:, :, 1:21, 0:26].double().sum(dim=(0, 2, 3))
    assert torch.allclose(out.cpu().double(), ref, rtol=1e-5, atol=1e-4)


def test_conv3x3_dgrad_fused_bn_bwd_bf16():
    """dX launch with pass 1 of the input unit's BatchNorm+ReLU backward in the epilogue == plain dX followed by
    gsd_bf16_bn_bwd_reduce(mode 0): same dz (bit for bit) and the same per-channel sums."""
    L = _lib()
    g = torch.Generator().manual_seed(14)
    n, h, w, cin, cout = 2, 19, 37, 96, 64          # dX has M = cin = 96 rows
    wt = torch.randn((cout, cin, 3, 3), generator=g) / (3.0 * cin ** 0.5)
    dy = bf16r(torch.randn((n, cout, h, w), generator=g))
    y = bf16r(torch.randn((n, cin, h, w), generator=g))
    mean, var = y.mean(dim=(0, 2, 3)), y.var(dim=(0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    gamma, beta = torch.rand((cin,), generator=g) + 0.5, 0.3 * torch.randn((cin,), generator=g)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    dev = [t.cuda() for t in (scale, shift, mean, invstd)]
    img = torch.empty((L.lib.gsd_bf16_weight_image_size(1, cout, cin),), dtype=torch.bfloat16, device="cuda")
    wd = wt.cuda()
    L.check(L.lib.gsd_bf16_weight_image(1, wd.data_ptr(), cout, cin, img.data_ptr(), L.stream_ptr()), "wimg")
    dyb, yb = to_nhwc(dy), to_nhwc(y)
    din, dyv = L.make_nhwc(dyb), L.make_nhwc(yb)
    # (a) unfused
    dx = torch.zeros((n, h, w, cin), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib.gsd_bf16_conv3x3(C.byref(din), img.data_ptr(), C.byref(L.make_nhwc(dx)), cout, cin, None, None, L.stream_ptr()), "dX")
    rows_b = L.lib.gsd_bf16_bn_bwd_partial_rows(n, h, w)
    part_b = torch.zeros((rows_b, 3 * cin), device="cuda")
    L.check(L.lib.gsd_bf16_bn_bwd_reduce(0, C.byref(dyv), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(),
                                         C.byref(L.make_nhwc(dx)), C.byref(dyv), C.byref(dyv), None, None, C.byref(L.make_nhwc(dx)),
                                         part_b.data_ptr(), L.stream_ptr()), "reduce")
    # (b) fused
    dz = torch.zeros((n, h, w, cin), dtype=torch.bfloat16, device="cuda")
    rows = L.lib.gsd_bf16_conv_partial_rows(n, h, w, cin)
    mp = L.lib.gsd_bf16_conv_mpad(cin)
    part = torch.zeros((rows, 2 * mp), device="cuda")
    bw = L.gsd_bf16_bnbwd()
    bw.y = C.pointer(dyv)
    bw.scale, bw.shift, bw.mean, bw.invstd = (t.data_ptr() for t in dev)
    L.check(L.lib.gsd_bf16_conv3x3(C.byref(din), img.data_ptr(), C.byref(L.make_nhwc(dz)), cout, cin, part.data_ptr(), C.byref(bw),
                                   L.stream_ptr()), "fused dX")
    torch.cuda.synchronize()
    assert torch.equal(dz.cpu(), dx.cpu())
    sa, sb = part.double().sum(dim=0).cpu(), part_b.double().sum(dim=0).cpu()
    assert torch.allclose(sa[:cin], sb[:cin], rtol=1e-5, atol=1e-4)
    assert torch.allclose(sa[mp:mp + cin], sb[cin:2 * cin], rtol=1e-5, atol=1e-4)


def test_conv3x3_bnrelu_fused_eval_bf16():
    """Eval-mode conv + BatchNorm + ReLU in one kernel == conv, then gsd_bf16_bn_apply, up to one rounding (the fused
    form never rounds the raw conv output to bf16)."""
    L = _lib()
    g = torch.Generator().manual_seed(15)
    n, h, w, k, m = 2, 18, 29, 64, 96
    x = bf16r(torch.randn((n, k, h, w), generator=g))
    wt = bf16r(torch.randn((m, k, 3, 3), generator=g) / (3.0 * k ** 0.5))
    scale, shift = torch.rand((m,), generator=g) + 0.5, 0.3 * torch.randn((m,), generator=g)
    ref = torch.relu(F.conv2d(x.double(), wt.double(), padding=1) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None])
    xin, out = to_nhwc(x), torch.zeros((n, h, w, m), dtype=torch.bfloat16, device="cuda")
    img = weight_image(wt.permute(2, 3, 0, 1).reshape(9, m, k))
    sc_d, sh_d = scale.cuda(), shift.cuda()
    L.check(L.lib.gsd_bf16_conv3x3_bnrelu(C.byref(L.make_nhwc(xin)), img.data_ptr(), C.byref(L.make_nhwc(out)), k, m, sc_d.data_ptr(),
                                          sh_d.data_ptr(), L.stream_ptr()), "conv_bnrelu")
    assert_close_bf16(from_nhwc(out, 0, m), ref, "conv3x3+bn+relu")


@pytest.mark.parametrize("n,c,h,w,m", [(2, 3, 21, 27, 32), (1, 3, 37, 130, 64), (2, 1, 9, 70, 32), (1, 2, 5, 64, 64), (3, 3, 4, 3, 64)])
def test_first_layer_direct_kernels_bf16(n, c, h, w, m):
    """gsd_bf16_conv3x3_first / gsd_bf16_wgrad_first (the first layer straight from x) against the im2col path they replace:
    the forward multiplies the same bf16 operands in the same k order through the same MFMA -- bit-identical raw output, equal
    BatchNorm sums, bit-identical fused eval output; dW (with the BatchNorm backward applied on the fly) against
    gsd_bf16_bn_bwd_apply + im2col + gsd_bf16_wgrad and against fp64, twice the same bits."""
    L = _lib()
    g = torch.Generator().manual_seed(n * 100 + w + m)
    assert L.lib.gsd_bf16_conv3x3_first_supported(c, m) == 1 and L.lib.gsd_bf16_conv3x3_first_supported(4, m) == 0
    assert L.lib.gsd_bf16_conv3x3_first_supported(c, 48) == 0
    x = torch.rand((n, c, h, w), generator=g)
    wt = torch.randn((m, c, 3, 3), generator=g) * 0.3
    x_d, w_d = x.cuda(), wt.cuda()
    img = torch.zeros(L.lib.gsd_bf16_weight_image_size(2, m, c), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib.gsd_bf16_weight_image(2, w_d.data_ptr(), m, c, img.data_ptr(), L.stream_ptr()), "wimg")
    mp = L.lib.gsd_bf16_conv_mpad(m)
    # ---- forward: the im2col path
    col = torch.zeros((n, h, w, 32), dtype=torch.bfloat16, device="cuda")
    dcol = L.make_nhwc(col)
    L.check(L.lib.gsd_bf16_im2col3x3(x_d.data_ptr(), n, c, h, w, C.byref(dcol), L.stream_ptr()), "im2col")
    y0 = torch.zeros((n, h, w, m), dtype=torch.bfloat16, device="cuda")
    rows0 = L.lib.gsd_bf16_conv_partial_rows(n, h, w, m)
    part0 = torch.zeros((rows0, 2 * mp), device="cuda")
    z = L.int_array([0])
    d0 = L.make_nhwc(y0)
    L.check(L.lib.gsd_bf16_conv_dense(C.byref(dcol), img.data_ptr(), C.byref(d0), 32, m, 1, 1, z, z, h, w, 0, 0, 0, None,
                                      part0.data_ptr(), None, L.stream_ptr()), "dense")
    # ---- forward: straight from x
    y1 = torch.full((n, h, w, m), float("nan"), dtype=torch.bfloat16, device="cuda")
    rows1 = L.lib.gsd_bf16_conv3x3_first_partial_rows(n, h, w, m)
    part1 = torch.full((rows1, 2 * mp), float("nan"), device="cuda")
    d1 = L.make_nhwc(y1)
    L.check(L.lib.gsd_bf16_conv3x3_first(x_d.data_ptr(), n, c, h, w, img.data_ptr(), C.byref(d1), m, part1.data_ptr(), None, None,
                                         L.stream_ptr()), "first")
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    ref = F.conv2d(bf16r(x).double(), bf16r(wt).double(), padding=1)
    assert_close_bf16(from_nhwc(y1, 0, m), ref, "first layer")
    s0, s1 = part0.double().sum(0), part1.double().sum(0)
    yv = y1.double()
    np.testing.assert_allclose(s1[:m].cpu().numpy(), yv.sum(dim=(0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s1[mp:mp + m].cpu().numpy(), (yv * yv).sum(dim=(0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s1[:m].cpu().numpy(), s0[:m].cpu().numpy(), rtol=1e-5, atol=1e-3)
    # eval mode: BatchNorm (running statistics) + ReLU in the epilogue
    sc, sh = (torch.rand(m, generator=g) + 0.5).cuda(), (torch.randn(m, generator=g) * 0.2).cuda()
    a0 = torch.zeros((n, h, w, m), dtype=torch.bfloat16, device="cuda")
    a1 = torch.full((n, h, w, m), float("nan"), dtype=torch.bfloat16, device="cuda")
    da0, da1 = L.make_nhwc(a0), L.make_nhwc(a1)
    L.check(L.lib.gsd_bf16_conv1x1_bnrelu(C.byref(dcol), img.data_ptr(), C.byref(da0), 32, m, sc.data_ptr(), sh.data_ptr(),
                                          L.stream_ptr()), "1x1 bnrelu")
    L.check(L.lib.gsd_bf16_conv3x3_first(x_d.data_ptr(), n, c, h, w, img.data_ptr(), C.byref(da1), m, None, sc.data_ptr(),
                                         sh.data_ptr(), L.stream_ptr()), "first bnrelu")
    assert torch.equal(a0.view(torch.int16), a1.view(torch.int16))
    # ---- dW with the BatchNorm backward on the fly
    dz = bf16r(torch.randn((n, m, h, w), generator=g))
    vec = [(torch.rand(m, generator=g) + 0.5), torch.randn(m, generator=g) * 0.3, (torch.rand(m, generator=g) + 0.5),
           torch.randn(m, generator=g) * 0.1, torch.randn(m, generator=g) * 0.1]        # scale mean invstd c1 c2
    vd = [v.cuda() for v in vec]
    dz_d = to_nhwc(dz)
    need = L.lib.gsd_bf16_wgrad_first_workspace(n, h, w, m)
    ws = torch.zeros(need, device="cuda")
    outs = []
    ddz = L.make_nhwc(dz_d)
    for _ in range(2):
        dw = torch.full((m, c, 3, 3), float("nan"), device="cuda")
        L.check(L.lib.gsd_bf16_wgrad_first(x_d.data_ptr(), n, c, h, w, C.byref(ddz), C.byref(d1), *[v.data_ptr() for v in vd],
                                           dw.data_ptr(), ws.data_ptr(), need, L.stream_ptr()), "wgrad_first")
        outs.append(dw.cpu())
    assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1])
    # the path it replaces: apply pass (in place on a copy), then dW of the im2col'd input
    dr = dz_d.clone()
    ddr = L.make_nhwc(dr)
    L.check(L.lib.gsd_bf16_bn_bwd_apply(C.byref(ddr), C.byref(d1), vd[0].data_ptr(), vd[1].data_ptr(), vd[2].data_ptr(),
                                        vd[3].data_ptr(), vd[4].data_ptr(), L.stream_ptr()), "apply")
    old = _wgrad(L, dr, 0, m, col, 0, 32, 1, 1, [0], [0], 9 * c).reshape(m, c, 3, 3)
    scale_ref = float(old.abs().max())
    assert float((outs[0] - old).abs().max()) <= 2e-5 * scale_ref
    wz = torch.zeros((m, c, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv2d(bf16r(x).double(), wz, padding=1).backward(from_nhwc(dr, 0, m).double())
    assert float((outs[0].double() - wz.grad).abs().max()) <= 2e-5 * float(wz.grad.abs().max())
    # unfused form (y == NULL: the gradient already is d_raw) and the workspace check
    dw2 = torch.full((m, c, 3, 3), float("nan"), device="cuda")
    L.check(L.lib.gsd_bf16_wgrad_first(x_d.data_ptr(), n, c, h, w, C.byref(ddr), None, None, None, None, None, None,
                                       dw2.data_ptr(), ws.data_ptr(), need, L.stream_ptr()), "wgrad_first plain")
    assert float((dw2.cpu() - old).abs().max()) <= 2e-5 * scale_ref
    assert L.lib.gsd_bf16_wgrad_first(x_d.data_ptr(), n, c, h, w, C.byref(ddz), C.byref(d1), *[v.data_ptr() for v in vd],
                                      dw.data_ptr(), ws.data_ptr(), need - 1, L.stream_ptr()) == -4


@pytest.mark.parametrize("n,c,h,w", [(2, 3, 19, 70), (1, 3, 8, 64), (3, 3, 21, 27), (2, 1, 33, 130), (1, 2, 5, 3), (2, 3, 40, 427)])
def test_inc_block_fused_equals_its_unfused_kernels_bf16(n, c, h, w):
    """The `inc` double convolution without the first convolution's raw output in HBM (gsd_bf16_inc.hip; unet.py:7-20, :67)
    against the four unfused launches it replaces, fed the same statistics:
      * gsd_bf16_conv3x3_first(out = NULL): the same BatchNorm partial rows as the storing form, bit for bit;
      * gsd_bf16_inc_conv: a0 and y1 bit-identical to conv3x3_first -> bn_apply(relu) -> conv3x3 (tiles cut by the image edge,
        images narrower than a tile, a halo that crosses the image border on every side), y1's partial sums equal to the sums
        of the stored values;
      * gsd_bf16_first_bn_bwd_reduce: [sum dz | sum dz*xhat] of the masked gradient against fp64 on the stored y0;
      * gsd_bf16_wgrad_first_recompute: dW bit-identical to gsd_bf16_wgrad_first on (masked dz, stored y0)."""
    L = _lib()
    m = 64
    g = torch.Generator().manual_seed(7 * n + h + w)
    assert L.lib.gsd_bf16_inc_supported(c, m) == 1 and L.lib.gsd_bf16_inc_supported(c, 32) == 0 and L.lib.gsd_bf16_inc_supported(4, m) == 0
    x = torch.rand((n, c, h, w), generator=g)
    w0 = torch.randn((m, c, 3, 3), generator=g) * 0.3
    w1 = torch.randn((m, m, 3, 3), generator=g) * 0.05
    x_d, w0_d, w1_d = x.cuda(), w0.cuda(), w1.cuda()
    img0 = torch.zeros(L.lib.gsd_bf16_weight_image_size(2, m, c), dtype=torch.bfloat16, device="cuda")
    img1 = torch.zeros(L.lib.gsd_bf16_weight_image_size(0, m, m), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib.gsd_bf16_weight_image(2, w0_d.data_ptr(), m, c, img0.data_ptr(), L.stream_ptr()), "wimg0")
    L.check(L.lib.gsd_bf16_weight_image(0, w1_d.data_ptr(), m, m, img1.data_ptr(), L.stream_ptr()), "wimg1")
    mp = L.lib.gsd_bf16_conv_mpad(m)
    # ---- unfused forward
    y0 = torch.zeros((n, h, w, m), dtype=torch.bfloat16, device="cuda")
    rows0 = L.lib.gsd_bf16_conv3x3_first_partial_rows(n, h, w, m)
    part0 = torch.full((rows0, 2 * mp), float("nan"), device="cuda")
    dy0 = L.make_nhwc(y0)
    L.check(L.lib.gsd_bf16_conv3x3_first(x_d.data_ptr(), n, c, h, w, img0.data_ptr(), C.byref(dy0), m, part0.data_ptr(), None, None,
                                         L.stream_ptr()), "first")
    part0s = torch.full((rows0, 2 * mp), float("nan"), device="cuda")
    L.check(L.lib.gsd_bf16_conv3x3_first(x_d.data_ptr(), n, c, h, w, img0.data_ptr(), None, m, part0s.data_ptr(), None, None,
                                         L.stream_ptr()), "first, statistics only")
    assert torch.equal(part0[:, :m], part0s[:, :m]) and torch.equal(part0[:, mp:mp + m], part0s[:, mp:mp + m])
    assert L.lib.gsd_bf16_conv3x3_first(x_d.data_ptr(), n, c, h, w, img0.data_ptr(), None, m, None, None, None, L.stream_ptr()) != 0
    yv = y0.double()
    cnt = n * h * w
    mean = (yv.sum(dim=(0, 1, 2)) / cnt)
    var = (yv * yv).sum(dim=(0, 1, 2)) / cnt - mean * mean
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    gamma, beta = (torch.rand(m, generator=g) + 0.5).double().cuda(), (torch.randn(m, generator=g) * 0.3).double().cuda()
    scale, shift = (gamma * invstd).float(), (beta - mean * gamma * invstd).float()
    mean_f, invstd_f = mean.float(), invstd.float()
    cat = torch.full((n, h, w, m + 32), 7.0, dtype=torch.bfloat16, device="cuda")       # a0 as a channel slice of a wider buffer
    da_ref = L.make_nhwc(cat, 0, m)
    L.check(L.lib.gsd_bf16_bn_apply(C.byref(dy0), scale.data_ptr(), shift.data_ptr(), C.byref(da_ref), 1, L.stream_ptr()), "apply")
    y1_ref = torch.zeros((n, h, w, m), dtype=torch.bfloat16, device="cuda")
    rows1 = L.lib.gsd_bf16_conv_partial_rows(n, h, w, m)
    part1 = torch.zeros((rows1, 2 * mp), device="cuda")
    dy1r = L.make_nhwc(y1_ref)
    L.check(L.lib.gsd_bf16_conv3x3(C.byref(da_ref), img1.data_ptr(), C.byref(dy1r), m, m, part1.data_ptr(), None, L.stream_ptr()), "conv3x3")
    # ---- fused forward
    cat2 = torch.full((n, h, w, m + 32), 7.0, dtype=torch.bfloat16, device="cuda")
    y1 = torch.full((n, h, w, m), float("nan"), dtype=torch.bfloat16, device="cuda")
    rowsf = L.lib.gsd_bf16_inc_conv_partial_rows(n, h, w)
    partf = torch.full((rowsf, 2 * mp), float("nan"), device="cuda")
    da2, dy1 = L.make_nhwc(cat2, 0, m), L.make_nhwc(y1)
    L.check(L.lib.gsd_bf16_inc_conv(x_d.data_ptr(), n, c, h, w, img0.data_ptr(), scale.data_ptr(), shift.data_ptr(), img1.data_ptr(),
                                    C.byref(da2), C.byref(dy1), partf.data_ptr(), L.stream_ptr()), "inc_conv")
    torch.cuda.synchronize()
    assert torch.equal(cat.view(torch.int16), cat2.view(torch.int16)), "a0 (and the untouched rest of the buffer)"
    assert torch.equal(y1_ref.view(torch.int16), y1.view(torch.int16)), "y1"
    sf = partf.double().sum(0)
    y1v = y1.double()
    np.testing.assert_allclose(sf[:m].cpu().numpy(), y1v.sum(dim=(0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(sf[mp:mp + m].cpu().numpy(), (y1v * y1v).sum(dim=(0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    # ---- backward of the first layer without y0
    da = bf16r(torch.randn((n, m, h, w), generator=g))
    da_d = to_nhwc(da)
    dda = L.make_nhwc(da_d)
    partb = torch.full((rows0, 2 * mp), float("nan"), device="cuda")
    L.check(L.lib.gsd_bf16_first_bn_bwd_reduce(x_d.data_ptr(), n, c, h, w, img0.data_ptr(), C.byref(dda), scale.data_ptr(), shift.data_ptr(),
                                               mean_f.data_ptr(), invstd_f.data_ptr(), partb.data_ptr(), L.stream_ptr()), "first_bn_bwd_reduce")
    y0f = y0.float()
    mask = (y0f.double() * scale.double() + shift.double()) > 0     # the sign of fma(y, scale, shift): exact in fp64
    dz = torch.where(mask, da_d.float(), torch.zeros((), device="cuda"))
    xhat = (y0f - mean_f) * invstd_f
    sb = partb.double().sum(0)
    s1_ref, s2_ref = dz.double().sum(dim=(0, 1, 2)), (dz.double() * xhat.double()).sum(dim=(0, 1, 2))
    tol1 = 1e-5 * float(dz.double().abs().sum(dim=(0, 1, 2)).max()) + 1e-6
    assert float((sb[:m] - s1_ref).abs().max()) <= tol1
    assert float((sb[mp:mp + m] - s2_ref).abs().max()) <= 1e-5 * float((dz.double() * xhat.double()).abs().sum(dim=(0, 1, 2)).max()) + 1e-6
    c1, c2 = (torch.randn(m, generator=g) * 0.1).cuda(), (torch.randn(m, generator=g) * 0.1).cuda()
    need = L.lib.gsd_bf16_wgrad_first_workspace(n, h, w, m)
    ws = torch.zeros(need, device="cuda")
    dz_d = dz.to(torch.bfloat16).contiguous()
    ddz = L.make_nhwc(dz_d)
    dw_ref = torch.full((m, c, 3, 3), float("nan"), device="cuda")
    L.check(L.lib.gsd_bf16_wgrad_first(x_d.data_ptr(), n, c, h, w, C.byref(ddz), C.byref(dy0), scale.data_ptr(), mean_f.data_ptr(),
                                       invstd_f.data_ptr(), c1.data_ptr(), c2.data_ptr(), dw_ref.data_ptr(), ws.data_ptr(), need,
                                       L.stream_ptr()), "wgrad_first")
    dw = torch.full((m, c, 3, 3), float("nan"), device="cuda")
    L.check(L.lib.gsd_bf16_wgrad_first_recompute(x_d.data_ptr(), n, c, h, w, img0.data_ptr(), C.byref(dda), scale.data_ptr(),
                                                 shift.data_ptr(), mean_f.data_ptr(), invstd_f.data_ptr(), c1.data_ptr(), c2.data_ptr(),
                                                 dw.data_ptr(), ws.data_ptr(), need, L.stream_ptr()), "wgrad_first_recompute")
    assert bool(torch.isfinite(dw).all()) and torch.equal(dw, dw_ref)


@pytest.mark.parametrize("n,h,w", [(2, 19, 70), (1, 8, 64), (3, 21, 27), (2, 33, 130), (1, 5, 3), (2, 40, 427)])
def test_conv3x3_c64_weights_resident_equals_the_dma_kernel_bf16(n, h, w):
    """gsd_bf16_conv3x3_c64 (64 -> 64 channels, weights resident in LDS, one halo fill per tile) against gsd_bf16_conv3x3: the same
    products in the same order -- raw output bit-identical, BatchNorm partial sums equal to the sums of the stored values; with
    `bw` (fused pass 1 of the BatchNorm + ReLU backward, raw output of the unit below from HBM) dz bit-identical and the two sums
    equal to those of the DMA kernel; without partials a plain convolution.  Input and output as channel slices of wider buffers."""
    L = _lib()
    m = 64
    g = torch.Generator().manual_seed(3 * n + h + w)
    assert L.lib.gsd_bf16_conv3x3_c64_supported(64, 64) == 1 and L.lib.gsd_bf16_conv3x3_c64_supported(128, 64) == 0
    a = bf16r(torch.randn((n, m, h, w), generator=g))
    wt = torch.randn((m, m, 3, 3), generator=g) * 0.05
    a_buf = to_nhwc(a, c_total=m + 32, c_off=32)              # the input is a channel slice of a wider buffer
    w_d = wt.cuda()
    mp = L.lib.gsd_bf16_conv_mpad(m)
    rows_d, rows_c = L.lib.gsd_bf16_conv_partial_rows(n, h, w, m), L.lib.gsd_bf16_conv3x3_c64_partial_rows(n, h, w)
    din = L.make_nhwc(a_buf, 32, m)
    for mode in (0, 1):                                       # forward image, dX image
        img = torch.zeros(L.lib.gsd_bf16_weight_image_size(mode, m, m), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib.gsd_bf16_weight_image(mode, w_d.data_ptr(), m, m, img.data_ptr(), L.stream_ptr()), "wimg")
        ref = torch.full((n, h, w, m + 16), 7.0, dtype=torch.bfloat16, device="cuda")
        got = torch.full((n, h, w, m + 16), 7.0, dtype=torch.bfloat16, device="cuda")
        pr, pg = torch.zeros((rows_d, 2 * mp), device="cuda"), torch.full((rows_c, 2 * mp), float("nan"), device="cuda")
        L.check(L.lib.gsd_bf16_conv3x3(C.byref(din), img.data_ptr(), C.byref(L.make_nhwc(ref, 16, m)), m, m, pr.data_ptr(), None,
                                       L.stream_ptr()), "conv3x3")
        L.check(L.lib.gsd_bf16_conv3x3_c64(C.byref(din), img.data_ptr(), C.byref(L.make_nhwc(got, 16, m)), pg.data_ptr(), None,
                                           L.stream_ptr()), "conv3x3_c64")
        torch.cuda.synchronize()
        assert torch.equal(ref.view(torch.int16), got.view(torch.int16)), f"mode {mode}: raw output (and the untouched channels)"
        yv = got[..., 16:].double()
        sg = pg.double().sum(0)
        np.testing.assert_allclose(sg[:m].cpu().numpy(), yv.sum(dim=(0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(sg[mp:mp + m].cpu().numpy(), (yv * yv).sum(dim=(0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    # ---- fused BatchNorm + ReLU backward pass 1, raw output of the unit below from HBM
    yb = bf16r(torch.randn((n, m, h, w), generator=g))
    yb_buf = to_nhwc(yb)
    vec = [(torch.rand(m, generator=g) + 0.5).cuda(), (torch.randn(m, generator=g) * 0.3).cuda(), (torch.randn(m, generator=g) * 0.1).cuda(),
           (torch.rand(m, generator=g) + 0.5).cuda()]      # scale shift mean invstd
    yv_ = L.make_nhwc(yb_buf)
    bw = L.gsd_bf16_bnbwd()
    bw.y = C.pointer(yv_)
    bw.scale, bw.shift, bw.mean, bw.invstd = [v.data_ptr() for v in vec]
    ref = torch.zeros((n, h, w, m), dtype=torch.bfloat16, device="cuda")
    got = torch.full((n, h, w, m), float("nan"), dtype=torch.bfloat16, device="cuda")
    pr, pg = torch.zeros((rows_d, 2 * mp), device="cuda"), torch.full((rows_c, 2 * mp), float("nan"), device="cuda")
    L.check(L.lib.gsd_bf16_conv3x3(C.byref(din), img.data_ptr(), C.byref(L.make_nhwc(ref)), m, m, pr.data_ptr(), C.byref(bw), L.stream_ptr()),
            "conv3x3 bw")
    L.check(L.lib.gsd_bf16_conv3x3_c64(C.byref(din), img.data_ptr(), C.byref(L.make_nhwc(got)), pg.data_ptr(), C.byref(bw), L.stream_ptr()),
            "conv3x3_c64 bw")
    torch.cuda.synchronize()
    assert torch.equal(ref.view(torch.int16), got.view(torch.int16)), "dz"
    sr, sg = pr.double().sum(0), pg.double().sum(0)
    mag = float(ref.double().abs().sum(dim=(0, 1, 2)).max()) + 1e-9
    assert float((sr[:m] - sg[:m]).abs().max()) <= 1e-5 * mag and float((sr[mp:mp + m] - sg[mp:mp + m]).abs().max()) <= 2e-5 * mag
    # ---- a plain convolution (partials == NULL): what inc's second dX runs when the first raw output was never stored
    ref = torch.zeros((n, h, w, m), dtype=torch.bfloat16, device="cuda")
    got = torch.full((n, h, w, m), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib.gsd_bf16_conv3x3(C.byref(din), img.data_ptr(), C.byref(L.make_nhwc(ref)), m, m, None, None, L.stream_ptr()), "plain")
    L.check(L.lib.gsd_bf16_conv3x3_c64(C.byref(din), img.data_ptr(), C.byref(L.make_nhwc(got)), None, None, L.stream_ptr()), "plain c64")
    torch.cuda.synchronize()
    assert torch.equal(ref.view(torch.int16), got.view(torch.int16))


@pytest.mark.parametrize("n,h,w,c", [(2, 13, 18, 64), (3, 9, 11, 40), (1, 20, 26, 128)])
def test_output_conv_with_batchnorm_folded_in_equals_apply_then_conv_bf16(n, h, w, c):
    """gsd_bf16_bn_relu_conv1x1_out (the last unit's BatchNorm + ReLU formed inside the 1x1 output convolution, unet.py:15-16 then
    :54) == gsd_bf16_bn_apply followed by gsd_bf16_conv1x1_out, bit for bit, and both match torch fp64 on the bf16 activation."""
    L = _lib()
    g = torch.Generator().manual_seed(3 * c + h)
    y = bf16r(torch.randn((n, c, h, w), generator=g))
    scale, shift = torch.rand((c,), generator=g) + 0.5, 0.4 * torch.randn((c,), generator=g)
    wo, bo = torch.randn((1, c), generator=g) / c ** 0.5, torch.randn((1,), generator=g)
    dev = [t.float().cuda() for t in (scale, shift, wo, bo)]
    ybuf = to_nhwc(y)
    a = torch.zeros((n, h, w, c), dtype=torch.bfloat16, device="cuda")
    out0 = torch.full((n, 1, h, w), float("nan"), device="cuda")
    out1 = torch.full((n, 1, h, w), float("nan"), device="cuda")
    dy_ = L.make_nhwc(ybuf)
    L.check(L.lib.gsd_bf16_bn_apply(C.byref(dy_), dev[0].data_ptr(), dev[1].data_ptr(), C.byref(L.make_nhwc(a)), 1, L.stream_ptr()), "apply")
    L.check(L.lib.gsd_bf16_conv1x1_out(C.byref(L.make_nhwc(a)), dev[2].data_ptr(), dev[3].data_ptr(), 1, out0.data_ptr(), L.stream_ptr()), "out")
    L.check(L.lib.gsd_bf16_bn_relu_conv1x1_out(C.byref(dy_), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), 1,
                                               out1.data_ptr(), L.stream_ptr()), "bn_relu_out")
    torch.cuda.synchronize()
    assert torch.equal(out0, out1)
    ref = F.conv2d(from_nhwc(a, 0, c).double(), wo.double()[:, :, None, None], bo.double())
    assert float((out1.cpu().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("n,H,W,ctot,cup,oy,ox,hh,ww", [(2, 21, 27, 96, 32, 0, 1, 20, 26), (3, 16, 24, 64, 32, 0, 0, 16, 24),
                                                        (2, 19, 23, 128, 64, 1, 2, 16, 20)])
def test_convT_bias_gradient_from_dx_statistics_bf16(n, H, W, ctot, cup, oy, ox, hh, ww):
    """gsd_bf16_convT_bias_grad: per-channel sums over the transposed convolution's window of the gradient slice, assembled from
    the statistics rows of a gsd_bf16_conv3x3 launch that wrote the WHOLE concat gradient (sums of what it stored) minus the F.pad
    strips -- against gsd_bf16_channel_sums over the window and against torch fp64 on the stored bf16 values."""
    L = _lib()
    g = torch.Generator().manual_seed(H * 7 + ctot)
    k = 64
    x = bf16r(torch.randn((n, k, H, W), generator=g))
    wt = bf16r(torch.randn((ctot, k, 3, 3), generator=g) / (3.0 * k ** 0.5))
    xin = to_nhwc(x)
    img = weight_image(wt.permute(2, 3, 0, 1).reshape(9, ctot, k))
    gcat = torch.zeros((n, H, W, ctot), dtype=torch.bfloat16, device="cuda")
    rows = L.lib.gsd_bf16_conv_partial_rows(n, H, W, ctot)
    mp = L.lib.gsd_bf16_conv_mpad(ctot)
    part = torch.full((rows, 2 * mp), float("nan"), device="cuda")
    L.check(L.lib.gsd_bf16_conv3x3(C.byref(L.make_nhwc(xin)), img.data_ptr(), C.byref(L.make_nhwc(gcat)), k, ctot, part.data_ptr(), None,
                                   L.stream_ptr()), "conv3x3 + statistics")
    gup = L.make_nhwc(gcat, ctot - cup, cup)
    nws = max(L.lib.gsd_bf16_convT_bias_grad_workspace(n, H, W, oy, ox, hh, ww, cup), L.lib.gsd_bf16_channel_sums_workspace(n, hh, ww, cup))
    ws = torch.empty((nws,), device="cuda")
    db0 = torch.full((cup,), float("nan"), device="cuda")
    db1 = torch.full((cup,), float("nan"), device="cuda")
    L.check(L.lib.gsd_bf16_channel_sums(C.byref(gup), oy, ox, hh, ww, db0.data_ptr(), ws.data_ptr(), nws, L.stream_ptr()), "channel_sums")
    L.check(L.lib.gsd_bf16_convT_bias_grad(part.data_ptr(), rows, 2 * mp, ctot - cup, C.byref(gup), oy, ox, hh, ww, db1.data_ptr(),
                                           ws.data_ptr(), nws, L.stream_ptr()), "convT_bias_grad")
    torch.cuda.synchronize()
    ref = gcat.double().cpu()[:, oy:oy + hh, ox:ox + ww, ctot - cup:].sum(dim=(0, 1, 2))
    scale = float(gcat.double().cpu()[..., ctot - cup:].abs().sum(dim=(0, 1, 2)).max())
    for got in (db0, db1):
        assert float((got.cpu().double() - ref).abs().max()) <= 2e-6 * scale + 1e-9
