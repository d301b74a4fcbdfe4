"""K-slab form of the two Winograd conv3x3 kernels (gsd_conv3x3_w43_ws / gsd_conv3x3_w43_dgrad_bnrelu_ws and their
gsd_conv3x3_w2d_* twins, include/gsd.h): the launches of
the 40x53 / 20x26 levels at small per-GPU batches cut their input channels into S slabs and a second launch adds the slabs and
runs the epilogue.  Same operator as /root/reference/gelslim_depth/models/unet.py:11,14 (forward) and its dX.

Checked here, at the real layer shapes and batch 8 (configs[3]'s per-GPU share on 8 GPUs), for forced S = 2, 3, 5 and the
planner's own choice: every launch form the engine uses (deferred-BatchNorm source, pooled plain source, two-segment decoder
source with the slab boundary on / across / behind the segment switch, two cropped dX destinations with statistics, the fused
BatchNorm-backward dX epilogue) against (a) the unsplit launch of the same kernel -- identical arithmetic up to the order of
S partial sums over up to 9216 products, bound 1e-5 relative L1 (measured 2-4e-6 at K = 1024) -- and (b) oracle/unet_numpy.py on one image at the op tolerance 1e-5; the statistics
rows against fp64 sums of what was stored; run-to-run bitwise equality; and a too-small scratch buffer falling back to fewer
slabs instead of overrunning it.
"""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch

from conftest import rel_l1

pytestmark = pytest.mark.gpu

HS = [320, 160, 80, 40, 20]
WS = [427, 213, 106, 53, 26]
N = 8

# (name, level, skip/plain channels, up-sampled channels, Cout, input is a pooled (plain) tensor)
UNITS = [
    ("down2.c0", 3, 256, 0, 512, True),
    ("down2.c1|up0.c1", 3, 512, 0, 512, False),
    ("down3.c0", 4, 512, 0, 1024, True),
    ("down3.c1", 4, 1024, 0, 1024, False),
    ("up0.c0", 3, 512, 512, 512, False),
    ("down1.c1", 2, 256, 0, 256, False),       # 80x106: unfolded tile grid, split only when forced
]
SPLITS = ["auto", "2", "3", "5"]
CASES = [(u, s) for u in UNITS for s in SPLITS if not (u[0] == "down1.c1" and s in ("auto", "5"))]


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


def rnd(rng, *shape, scale=1.0):
    return rng.standard_normal(shape, dtype=np.float32) * np.float32(scale)


def layout(gsd, mode, w, co, ci):
    wt = torch.zeros(gsd.lib.gsd_weight_layout_size(mode, co, ci), device="cuda")
    gsd.check(gsd.lib.gsd_weight_layout(mode, w.data_ptr(), co, ci, wt.data_ptr(), gsd.stream_ptr()))
    return wt


def bcast(v):
    return v[None, :, None, None]


def close(a, b, tol=1e-5):
    a64, b64 = a.double(), b.double()
    return float((a64 - b64).abs().sum() / b64.abs().sum().clamp_min(1e-30)) < tol


class _Form:
    """The entry points of one Winograd form by name: row form (w43, weight layouts 4 / 5) or two-dimensional form (w2d, 8 / 9)."""

    def __init__(self, L, name):
        self.name = name
        self.env = "GSD_W43_SPLIT" if name == "w43" else "GSD_W2D_SPLIT"
        self.mode_f, self.mode_d = (4, 5) if name == "w43" else (8, 9)
        for f in ("", "_ws", "_dgrad_bnrelu", "_dgrad_bnrelu_ws", "_workspace", "_partial_rows"):
            setattr(self, "conv" + f, getattr(L, f"gsd_conv3x3_{name}{f}"))


@pytest.mark.parametrize("form", ["w43", "w2d"])
@pytest.mark.parametrize("unit,split", CASES, ids=[f"{u[0]}-S{s}" for u, s in CASES])
def test_kslab_launches_match_unsplit_and_oracle(gsd, monkeypatch, unit, split, form):
    from oracle import unet_numpy as on
    name, lvl, c0, c1, co, pooled = unit
    h, w = HS[lvl], WS[lvl]
    ci = c0 + c1
    L = gsd.lib
    F = _Form(L, form)
    n = N
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 10000 + 77)
    monkeypatch.delenv(F.env, raising=False)
    if split != "auto":
        monkeypatch.setenv(F.env, split)
    need_f = F.conv_workspace(n, h, w, ci, co)
    need_d = F.conv_workspace(n, h, w, co, ci)
    if split == "auto" and lvl >= 3 and form == "w43":
        assert need_f > 0 or need_d > 0, "the planner splits the deep levels at batch 8"
    if split != "auto":
        assert need_f > 0 and need_d > 0, "a forced slab count applies wherever the shape admits it"
    ws = torch.full((max(need_f, need_d, 1) + 64,), float("nan"), device="cuda")
    guard = ws[max(need_f, need_d, 1):]                      # never written

    raw0 = rnd(rng, n, c0, h, w)
    if pooled:
        a0 = raw0
        r0d = slack_dev(gsd, raw0)
        segs = [gsd.make_src(r0d, slack=gsd.SLACK)]
    else:
        sc, sh = rng.uniform(0.5, 1.5, c0).astype(np.float32), rnd(rng, c0, scale=0.3)
        a0 = np.maximum(raw0 * bcast(sc) + bcast(sh), 0)
        r0d, scd, shd = slack_dev(gsd, raw0), dev(sc), dev(sh)
        segs = [gsd.make_src(r0d, scd, shd, relu=True, slack=gsd.SLACK)]
    top = left = 0
    if c1:
        uh, uw = 2 * HS[lvl + 1], 2 * WS[lvl + 1]
        up = rnd(rng, n, c1, uh, uw)
        upp, (top, left) = on.pad_to(up[:1], h, w)
        a_img0 = np.concatenate([a0[:1], upp], 1)
        upd = slack_dev(gsd, up)
        segs.append(gsd.make_src(upd, off=(top, left), slack=gsd.SLACK))
    else:
        a_img0 = a0[:1]
    wt_ = rnd(rng, co, ci, 3, 3, scale=1.0 / np.sqrt(9 * ci))
    wd = dev(wt_)
    src = gsd.src_array(segs)
    st = gsd.stream_ptr()
    mpad_o, mpad_i = (co + 63) // 64 * 64, (ci + 63) // 64 * 64

    # ---- forward + statistics: unsplit, split, split again
    wl_f = layout(gsd, F.mode_f, wd, co, ci)
    rows = F.conv_partial_rows(n, h, w, co)

    def fwd(use_ws):
        y = torch.full((n, co, h, w), float("nan"), device="cuda")
        part = torch.zeros(rows * 2 * mpad_o, device="cuda")
        dst = gsd.dst_array([gsd.make_dst(y)])
        if use_ws:
            gsd.check(F.conv_ws(src, len(segs), wl_f.data_ptr(), ci, co, dst, 1, part.data_ptr(), ws.data_ptr(), need_f, n, h, w, st))
        else:
            gsd.check(F.conv(src, len(segs), wl_f.data_ptr(), ci, co, dst, 1, part.data_ptr(), n, h, w, st))
        return y, part
    y0, p0 = fwd(False)
    y1, p1 = fwd(True)
    y2, p2 = fwd(True)
    assert bool(torch.isfinite(y1).all())
    assert close(y1, y0)
    assert torch.equal(y1, y2) and torch.equal(p1, p2), "run-to-run bitwise"
    if need_f > 0:
        assert not torch.equal(y1, y0), "the slab form did run (two partial sums round differently from one)"
    assert rel_l1(y1[:1].cpu().numpy(), on.conv3x3_fwd(a_img0, wt_)) < 1e-5
    sums = torch.zeros(65 * 2 * co, device="cuda", dtype=torch.float64)
    gsd.check(L.gsd_bn_reduce_partials(p1.data_ptr(), rows, mpad_o, co, sums.data_ptr(), st))
    y64 = y1.double()
    np.testing.assert_allclose(sums[:co].cpu().numpy(), y64.sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(sums[co:2 * co].cpu().numpy(), (y64 * y64).sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-5)
    del y0, y1, y2, y64

    # ---- dX in the engine's form for this unit
    dy = rnd(rng, n, co, h, w)
    dyp = pitched(dev(dy))
    wl_d = layout(gsd, F.mode_d, wd, co, ci)
    dsrc = gsd.src_array([gsd.make_src(dyp)])
    rows_d = F.conv_partial_rows(n, h, w, ci)
    dxr, _ = on.conv3x3_bwd(a_img0, wt_, dy[:1])
    if c1:
        def dx(use_ws):
            g_skip = torch.full((n, c0, h, w), float("nan"), device="cuda")
            g_up = torch.full((n, c1, uh, uw), float("nan"), device="cuda")
            part = torch.zeros(rows_d * 2 * mpad_i, device="cuda")
            dst = gsd.dst_array([gsd.make_dst(g_skip), gsd.make_dst(g_up, off=(top, left))])
            if use_ws:
                gsd.check(F.conv_ws(dsrc, 1, wl_d.data_ptr(), co, ci, dst, 2, part.data_ptr(), ws.data_ptr(), need_d, n, h, w, st))
            else:
                gsd.check(F.conv(dsrc, 1, wl_d.data_ptr(), co, ci, dst, 2, part.data_ptr(), n, h, w, st))
            return g_skip, g_up, part
        a_, b_, p_ = dx(False)
        a1, b1, p1 = dx(True)
        a2, b2, p2 = dx(True)
        assert close(a1, a_) and close(b1, b_)
        assert torch.equal(a1, a2) and torch.equal(b1, b2) and torch.equal(p1, p2)
        assert rel_l1(a1[:1].cpu().numpy(), dxr[:, :c0]) < 1e-5
        assert rel_l1(b1[:1].cpu().numpy(), dxr[:, c0:, top:top + uh, left:left + uw]) < 1e-5
        sums_d = torch.zeros(65 * 2 * ci, device="cuda", dtype=torch.float64)
        gsd.check(L.gsd_bn_reduce_partials(p1.data_ptr(), rows_d, mpad_i, ci, sums_d.data_ptr(), st))
        np.testing.assert_allclose(sums_d[c0:ci].cpu().numpy(), b1.double().sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-4, atol=1e-2)
    elif pooled:
        def dx(use_ws):
            g = torch.full((n, ci, h, w), float("nan"), device="cuda")
            dst = gsd.dst_array([gsd.make_dst(g)])
            if use_ws:
                gsd.check(F.conv_ws(dsrc, 1, wl_d.data_ptr(), co, ci, dst, 1, None, ws.data_ptr(), need_d, n, h, w, st))
            else:
                gsd.check(F.conv(dsrc, 1, wl_d.data_ptr(), co, ci, dst, 1, None, n, h, w, st))
            return g
        g0, g1, g2 = dx(False), dx(True), dx(True)
        assert close(g1, g0) and torch.equal(g1, g2)
        assert rel_l1(g1[:1].cpu().numpy(), dxr) < 1e-5
    else:
        mean, invstd = rnd(rng, c0, scale=0.3), rng.uniform(0.5, 2.0, c0).astype(np.float32)
        vecs = [scd, shd, dev(mean), dev(invstd)]
        s_, = [gsd.make_src(dyp)]

        def dx(use_ws):
            dz = torch.full((n, ci, h, w), float("nan"), device="cuda")
            part = torch.zeros(rows_d * 2 * mpad_i, device="cuda")
            d = gsd.make_dst(dz)
            if use_ws:
                gsd.check(F.conv_dgrad_bnrelu_ws(C.byref(s_), wl_d.data_ptr(), co, ci, C.byref(d), r0d.data_ptr(),
                                                 *[v.data_ptr() for v in vecs], part.data_ptr(), ws.data_ptr(), need_d, n, h, w, st))
            else:
                gsd.check(F.conv_dgrad_bnrelu(C.byref(s_), wl_d.data_ptr(), co, ci, C.byref(d), r0d.data_ptr(),
                                              *[v.data_ptr() for v in vecs], part.data_ptr(), n, h, w, st))
            return dz, part
        z0, q0 = dx(False)
        z1, q1 = dx(True)
        z2, q2 = dx(True)
        assert close(z1, z0) and torch.equal(z1, z2) and torch.equal(q1, q2)
        assert rel_l1(z1[:1].cpu().numpy(), dxr * (a0[:1] > 0)) < 1e-5
        sums_d = torch.zeros(65 * 2 * ci, device="cuda", dtype=torch.float64)
        gsd.check(L.gsd_bn_reduce_partials(q1.data_ptr(), rows_d, mpad_i, ci, sums_d.data_ptr(), st))
        z64 = z1.double()
        xhat = (r0d.double() - vecs[2].double()[None, :, None, None]) * vecs[3].double()[None, :, None, None]
        np.testing.assert_allclose(sums_d[:ci].cpu().numpy(), z64.sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-4, atol=1e-2)
        np.testing.assert_allclose(sums_d[ci:2 * ci].cpu().numpy(), (z64 * xhat).sum(dim=(0, 2, 3)).cpu().numpy(), rtol=1e-4, atol=1e-2)
    assert bool(torch.isnan(guard).all()), "nothing was written behind the scratch the library asked for"


@pytest.mark.parametrize("form", ["w43", "w2d"])
def test_kslab_scratch_too_small_falls_back(gsd, monkeypatch, form):
    """The launcher takes the scratch capacity and shrinks the slab count to what fits (down to the plain launch): a tuning
    switch changed between allocation and launch can never overrun the buffer."""
    L = gsd.lib
    F = _Form(L, form)
    n, h, w, ci, co = 8, 20, 26, 1024, 1024
    monkeypatch.setenv(F.env, "2")
    need2 = F.conv_workspace(n, h, w, ci, co)
    monkeypatch.setenv(F.env, "5")
    need5 = F.conv_workspace(n, h, w, ci, co)
    assert need5 * 2 == need2 * 5 and need2 > 0
    rng = np.random.default_rng(5)
    x = slack_dev(gsd, rnd(rng, n, ci, h, w))
    wd = dev(rnd(rng, co, ci, 3, 3, scale=0.01))
    wl = layout(gsd, F.mode_f, wd, co, ci)
    src = gsd.src_array([gsd.make_src(x, slack=gsd.SLACK)])
    outs = []
    for cap in (need5, need2 + 8, need2 - 4, 0):        # 5 slabs, 2 slabs (all that fits), none, none
        ws = torch.full((need5 + 64,), float("nan"), device="cuda")
        y = torch.full((n, co, h, w), float("nan"), device="cuda")
        gsd.check(F.conv_ws(src, 1, wl.data_ptr(), ci, co, gsd.dst_array([gsd.make_dst(y)]), 1, None,
                            ws.data_ptr() if cap else None, cap, n, h, w, gsd.stream_ptr()))
        assert bool(torch.isfinite(y).all())
        assert bool(torch.isnan(ws[cap:]).all()), "no write behind the stated capacity"
        outs.append(y)
    monkeypatch.setenv(F.env, "0")
    y = torch.empty_like(outs[0])
    gsd.check(F.conv(src, 1, wl.data_ptr(), ci, co, gsd.dst_array([gsd.make_dst(y)]), 1, None, n, h, w, gsd.stream_ptr()))
    assert torch.equal(outs[2], y) and torch.equal(outs[3], y), "no room for two slabs: the plain launch"
    assert close(outs[0], y) and close(outs[1], y) and not torch.equal(outs[0], outs[1])


def test_eval_forward_never_splits_and_train_step_is_reproducible(gsd):
    """Engine level: the eval-mode forward passes no scratch (image i of a batch == the image alone, bit for bit, whatever
    the planner would do at that batch); a batch-8 train step through the slab launches is run-to-run bitwise and its loss
    agrees with the unsplit step."""
    import os
    from gelslim_depth_amd import synth
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    dims = [64, 128, 256, 512, 1024]
    st = synth.make_state(3, 1, dims, 3, "conditioned")
    x, t = synth.make_batch(8, 160, 213, 4)          # levels: 160x213 ... 10x13 -- the three deepest fold and split
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()

    def model():
        m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
        return m.cuda()
    m = model().eval()
    with torch.no_grad():
        yb = m(x=xd).clone()
        y1 = m(x=xd[5:6].contiguous())
    assert torch.equal(yb[5:6], y1)
    res = []
    for env in (None, None, "0"):
        for key in ("GSD_W43_SPLIT", "GSD_W2D_SPLIT"):
            if env is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = env
        try:
            m = model().train()
            step = TrainStep(m)
            loss = float(step(xd, td))
            torch.cuda.synchronize()
            res.append((loss, step.g_flat.clone(), m._engine.conv_ws is not None))
        finally:
            os.environ.pop("GSD_W43_SPLIT", None)
            os.environ.pop("GSD_W2D_SPLIT", None)
    assert res[0][2] and not res[2][2], "slab scratch exists exactly when the planner splits some launch"
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
    assert abs(res[0][0] - res[2][0]) < 1e-5 * abs(res[2][0])
    g0, g2 = res[0][1].double(), res[2][1].double()
    assert float((g0 - g2).abs().sum() / g2.abs().sum()) < 5e-2      # chaotic in the last bits, same bound as the two conv forms
