"""GPU: the hardened edges of the boundary -- the non-finite guard (the device-side form of the reference's
`if pred_loss.isnan()` test, train_utils/train_unet.py:371-372), stale-forward detection of the one-node autograd
function, target dtype handling, and property tests of the EMA kernel that do not lean on the oracle's restatement of
torch_ema (train_unet.py:309,376; "parity unpinned")."""
import numpy as np
import pytest
import torch

from gelslim_depth_amd import synth

pytestmark = pytest.mark.gpu

DIMS = [8, 16, 32]


def make(nan_policy=None, lr=1e-3, wd=1e-6, seed=5, **kw):
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    st = synth.make_state(3, 1, DIMS, seed, "conditioned")
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=DIMS)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to("cuda").train()
    return m, TrainStep(m, lr=lr, weight_decay=wd, nan_policy=nan_policy, **kw)


def snapshot(m, step):
    d = {"p": step.p_flat.clone(), "m": step.m_flat.clone(), "v": step.v_flat.clone(), "ema": step.ema_flat.clone()}
    for k, v in m.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            d[k] = v.clone()
    return d


def same(a, b, keys=None):
    return all(torch.equal(a[k], b[k]) for k in (keys or a.keys()))


@pytest.mark.parametrize("where", ["target", "input"])
def test_nan_policy_skip_leaves_the_state_untouched(where):
    m, step = make("skip")
    x, t = synth.make_batch(2, 21, 27, 6)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    step(xd, td)                                   # a clean step first: moments and EMA are non-trivial
    assert step.skipped_steps() == 0
    before = snapshot(m, step)
    xb, tb = xd.clone(), td.clone()
    (tb if where == "target" else xb)[1, 0, 3, 4] = float("nan")
    loss = step(xb, tb)
    assert step.skipped_steps() == 1
    after = snapshot(m, step)
    if where == "target":                          # the forward was healthy: only the optimiser state is protected
        assert not np.isfinite(float(loss))
        assert same(before, after, ["p", "m", "v", "ema"])
    else:                                          # NaN batch statistics: they never reach the running statistics either
        assert same(before, after)
    for k, v in after.items():
        assert bool(torch.isfinite(v).all()), k
    step(xd, td)                                   # training carries on
    assert step.skipped_steps() == 1
    assert not torch.equal(step.p_flat, before["p"]) and bool(torch.isfinite(step.p_flat).all())
    assert np.isfinite(float(step.last_loss))


def test_nan_policy_raise_reports_at_the_check_and_none_is_unguarded():
    from gelslim_depth_amd._lib import GsdError
    m, step = make("raise")
    x, t = synth.make_batch(2, 21, 27, 6)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    step(xd, td)
    step.check_finite()                            # nothing to report
    p = step.p_flat.clone()
    tb = td.clone()
    tb[0, 0, 0, 0] = float("inf")
    step(xd, tb)
    assert torch.equal(step.p_flat, p)
    with pytest.raises(GsdError, match="non-finite"):
        step.check_finite()
    step.check_finite()                            # reported once
    # the reference's arithmetic, unguarded (nan_policy=None): the NaN gradient reaches the parameters
    m2, step2 = make(None)
    step2(xd, tb)
    assert step2.guard_words is None and not bool(torch.isfinite(step2.p_flat).all())


def test_backward_of_a_stale_forward_raises():
    """The activations live in the engine, not in the autograd context: two train-mode forwards before one backward
    (gradient accumulation, or a forward on another batch in between) must raise instead of back-propagating the wrong
    activations."""
    m, _ = make()
    x, t = synth.make_batch(2, 21, 27, 6)
    xd = torch.from_numpy(x).cuda()
    y1 = m(x=xd)
    y2 = m(x=xd * 0.5)
    with pytest.raises(RuntimeError, match="saved activations are gone"):
        y1.sum().backward()
    y2.sum().backward()                            # the latest forward is fine
    y3 = m(x=xd)
    m.eval()
    with torch.no_grad():
        m(x=xd)                                    # an eval pass in between also overwrites the buffers
    m.train()
    with pytest.raises(RuntimeError, match="saved activations are gone"):
        y3.sum().backward()


def test_gradient_accumulation_over_forward_backward_pairs():
    """What the one-node autograd function does support: accumulation over forward/backward PAIRS (the usual micro-batch loop:
    `for mb: loss(model(mb)).backward()`, then one optimiser step).  .grad after two pairs is the bitwise sum of the two
    gradients taken alone (autograd adds the second backward's result to .grad), on the plain module (no TrainStep arena)."""
    from gelslim_depth_amd.models.unet import UNet
    st = synth.make_state(3, 1, DIMS, 9, "conditioned")

    def fresh():
        m = UNet(n_channels=3, n_classes=1, layer_dimensions=DIMS)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
        return m.to("cuda").train()
    xa, ta = synth.make_batch(2, 21, 27, 6)
    xb, tb = synth.make_batch(2, 21, 27, 7)
    batches = [(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()) for x, t in ((xa, ta), (xb, tb))]
    alone = []
    for x, t in batches:
        m = fresh()
        torch.mean((m(x=x) - t) ** 2).backward()
        alone.append({k: p.grad.clone() for k, p in m.named_parameters()})
    m = fresh()
    for x, t in batches:
        torch.mean((m(x=x) - t) ** 2).backward()
    for k, p in m.named_parameters():
        assert torch.equal(p.grad, alone[0][k] + alone[1][k]), k


def test_target_dtype_is_cast_not_reinterpreted():
    from gelslim_depth_amd._lib import GsdError
    from gelslim_depth_amd.train import loss_fwd_bwd
    x, t = synth.make_batch(2, 21, 27, 6)
    xd = torch.from_numpy(x).cuda()
    losses = []
    for tt in (torch.from_numpy(t).cuda(), torch.from_numpy(t.astype(np.float64)).cuda()):
        _, step = make()
        losses.append(float(step(xd, tt)))
    assert losses[0] == losses[1]
    out = torch.zeros((2, 1, 4, 4), device="cuda")
    buf, ws = torch.zeros(1, device="cuda"), torch.zeros(2048, device="cuda", dtype=torch.float64)
    with pytest.raises(GsdError, match="float32"):
        loss_fwd_bwd("mse", out, out.double(), None, buf, ws)
    with pytest.raises(GsdError, match="shape"):
        loss_fwd_bwd("mse", out, out[:1], None, buf, ws)


# hand-computed torch_ema schedule d(n) = min(decay, (1+n)/(10+n)), decay 0.995:  n -> d
EMA_TABLE = {1: 2 / 11, 2: 3 / 12, 3: 4 / 13, 10: 11 / 20, 100: 101 / 110, 1000: 1001 / 1010,
             1790: 0.995, 1791: 0.995, 5000: 0.995}


def test_ema_schedule_of_the_step_matches_a_hand_computed_table():
    """With lr = 0 the parameters never move, so 1 - shadow follows  prod_k d(k)  exactly as the published rule says;
    the table is written out by hand (no oracle involved).  Checks TrainStep's host-side schedule AND the kernel."""
    m, step = make(lr=0.0, wd=0.0)
    x, t = synth.make_batch(1, 16, 16, 3)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    step.p_flat.fill_(1.0)
    step.ema_flat.zero_()
    gap = 1.0
    for n in range(1, 13):
        step(xd, td)
        d = min(0.995, (1 + n) / (10 + n))
        if n in EMA_TABLE:
            assert abs(d - EMA_TABLE[n]) < 1e-15
        gap *= d
        got = 1.0 - step.ema_flat.double()
        assert float(got.max() - got.min()) == 0.0          # every element took the same path
        assert abs(float(got[0]) - gap) <= 4e-7 * n, (n, float(got[0]), gap)
        assert bool((step.p_flat == 1.0).all())
    for n, d in EMA_TABLE.items():
        assert abs(min(0.995, (1.0 + n) / (10.0 + n)) - d) < 1e-15


def test_ema_kernel_properties():
    """Properties of shadow <- shadow - (1-d)(shadow - p) on the HIP kernel itself (gsd_adam_ema with lr = 0):
    fixed point when shadow == p, d = 1 freezes the shadow, the update is a contraction towards p that never overshoots,
    and it equals the convex combination d*shadow + (1-d)*p to rounding."""
    from gelslim_depth_amd import _lib as L
    rng = np.random.default_rng(3)
    n = 4099
    p = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    g = torch.zeros(n, device="cuda")
    mm, vv = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")

    def ema(shadow, d):
        s = shadow.clone()
        pp = p.clone()
        L.check(L.lib.gsd_adam_ema(pp.data_ptr(), g.data_ptr(), mm.data_ptr(), vv.data_ptr(), s.data_ptr(), n, 1, 0.0, 0.9,
                                   0.999, 1e-8, 0.0, d, 1.0, None, L.stream_ptr()), "adam_ema")
        assert torch.equal(pp, p)
        return s
    assert torch.equal(ema(p, 0.3), p)                           # fixed point
    s0 = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
    assert torch.equal(ema(s0, 1.0), s0)                         # d = 1: frozen
    prev = s0
    for d in (2 / 11, 0.25, 0.55, 0.995):
        s1 = ema(prev, d)
        assert bool(((s1 - p).abs() <= (prev - p).abs() * (1 + 1e-6) + 1e-7).all())                  # contraction
        assert bool((((prev - p) * (s1 - p)) >= -1e-12).all())                                       # no overshoot
        ref = d * prev.double() + (1 - d) * p.double()
        assert float((s1.double() - ref).abs().max()) <= 4e-7 * float(ref.abs().max() + 1)          # convex combination
        prev = s1


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_fused_train_step_dispatches_no_aten_compute_op(precision):
    """"No torch op computes any part of the path" (engine.py): every ATen operator that reaches the dispatcher during a
    fused train step (forward + loss + backward + Adam + EMA, after the buffers exist) must be an allocation or a view --
    the arithmetic, the BatchNorm counters and the ConvT bias gradients included, is libgsd launches."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    dims = [32, 64, 128]
    st = synth.make_state(3, 1, dims, 5, "conditioned")
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision=precision)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to("cuda").train()
    step = TrainStep(m)
    x, t = synth.make_batch(2, 37, 53, 6)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    step(xd, td)                                   # first call allocates the activation buffers (zero fills allowed there)
    nbt0 = {k: int(v) for k, v in m.state_dict().items() if k.endswith("num_batches_tracked")}
    seen = []

    class Recorder(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            seen.append(str(func))
            return func(*args, **(kwargs or {}))

    with Recorder():
        step(xd, td)
    torch.cuda.synchronize()
    harmless = ("aten.empty", "aten.view", "aten._unsafe_view", "aten.slice", "aten.select", "aten.as_strided", "aten.detach",
                "aten.alias", "aten.reshape", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.t.", "aten.permute",
                "aten.narrow", "aten.lift_fresh", "aten.new_empty", "aten.empty_like", "aten.empty_strided")
    computing = sorted({f for f in seen if not f.startswith(harmless)})
    assert computing == [], computing
    assert all(int(v) == nbt0[k] + 1 for k, v in m.state_dict().items() if k.endswith("num_batches_tracked"))


def test_partials_channel_sums_and_add_counters():
    """gsd_partials_channel_sums == the fp64 column sums of gsd_bn_reduce_partials' first half, rounded to fp32, for a channel
    range; gsd_add_counters bumps every int64 counter it is given (more than one launch's worth of them)."""
    from gelslim_depth_amd import _lib as gsd
    rng = np.random.default_rng(3)
    rows, mpad, c = 517, 128, 100
    part = torch.from_numpy(rng.standard_normal((rows, 2 * mpad)).astype(np.float32)).cuda()
    sums = torch.zeros(65 * 2 * c, dtype=torch.float64, device="cuda")
    out = torch.full((40,), float("nan"), device="cuda")
    gsd.check(gsd.lib.gsd_partials_channel_sums(part.data_ptr(), rows, mpad, c, 37, 40, out.data_ptr(), sums.data_ptr(),
                                                gsd.stream_ptr()))
    ref = part.double().sum(0)[:c]
    np.testing.assert_allclose(sums[:c].cpu().numpy(), ref.cpu().numpy(), rtol=1e-13, atol=1e-10)
    assert torch.equal(out, ref[37:77].float())
    assert gsd.lib.gsd_partials_channel_sums(part.data_ptr(), rows, mpad, c, 90, 40, out.data_ptr(), sums.data_ptr(),
                                             gsd.stream_ptr()) == -1
    counters = [torch.tensor(i, dtype=torch.int64, device="cuda") for i in range(70)]
    gsd.add_counters(counters, 3)
    assert [int(v) for v in counters] == [i + 3 for i in range(70)]
    with pytest.raises(gsd.GsdError):
        gsd.add_counters([torch.zeros((), dtype=torch.int32, device="cuda")])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_skipped_step_leaves_no_trace_in_any_running_statistic(precision):
    """One output channel of a MIDDLE layer goes NaN: the layers in front of it have finite batch statistics and have
    already updated their running statistics when the step is found bad, and so may the finite channels beside it.  With a
    nan_policy the step is bracketed by gsd_guard_snapshot / gsd_guard_restore: after the skipped step every running
    statistic, parameter, moment and EMA value is bit-identical to before -- on every rank alike."""
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep
    dims = [32, 64, 128] if precision == "bf16" else DIMS
    st = synth.make_state(3, 1, dims, 5, "conditioned")
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision=precision)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to("cuda").train()
    step = TrainStep(m, nan_policy="skip")
    x, t = synth.make_batch(2, 21, 27, 6)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    step(xd, td)
    w = dict(m.named_parameters())["down.0.maxpool_conv.1.double_conv.0.weight"]
    saved = w.data[3].clone()
    w.data[3] = float("nan")
    before = snapshot(m, step)
    nbt = {k: int(v) for k, v in m.state_dict().items() if k.endswith("num_batches_tracked")}
    step(xd, td)
    assert step.skipped_steps() == 1
    after = snapshot(m, step)
    for k in before:
        assert torch.equal(before[k].view(torch.int32), after[k].view(torch.int32)), k     # bitwise, NaN row included
    assert all(int(v) == nbt[k] + 1 for k, v in m.state_dict().items() if k.endswith("num_batches_tracked"))
    w.data[3] = saved
    step(xd, td)
    assert step.skipped_steps() == 1 and bool(torch.isfinite(step.p_flat).all())
    rm = m.state_dict()["inc.double_conv.1.running_mean"]
    assert not torch.equal(rm, before["inc.double_conv.1.running_mean"])          # a healthy step moves them again


def test_input_gradient_request_raises_instead_of_returning_none():
    """The reference module is differentiable w.r.t. its input (unet.py:79-88); libgsd builds no dX for the first convolution
    (nothing in the reference asks for it, train_unet.py:344-347): asking must raise, not hand back a silent None."""
    m, _ = make()
    x, _ = synth.make_batch(2, 21, 27, 6)
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    with pytest.raises(NotImplementedError, match="gradient with respect to its input"):
        m(x=xd)
    with torch.no_grad():
        m(x=xd)                                    # no graph asked for: fine
    m(x=xd.detach()).sum().backward()


@pytest.mark.parametrize("nan_policy", [None, "skip"])
def test_moving_the_model_after_trainstep_raises(nan_policy):
    """TrainStep re-points the module's parameters (and, with a nan_policy, its BatchNorm buffers) at flat arenas; a later
    model.to(...) / .double().float() re-allocates them and would leave the kernels updating arenas nobody reads."""
    from gelslim_depth_amd._lib import GsdError
    m, step = make(nan_policy=nan_policy)
    x, t = synth.make_batch(2, 21, 27, 6)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    step(xd, td)
    m.load_state_dict(m.state_dict())              # in place: the arenas stay
    step(xd, td)
    m.double().float()                             # re-allocates every tensor
    with pytest.raises(GsdError, match="no longer lives in the step's flat arena"):
        step(xd, td)
