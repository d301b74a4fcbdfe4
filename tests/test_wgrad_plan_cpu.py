"""Host logic of the conv3x3 dW dispatcher (gsd_conv3x3_wgrad_form / _mfma_count / _workspace in gsd_wgrad*.hip), on the CPU: the
queries read no device memory, so descriptors with made-up (aligned) addresses are enough.  What the engine relies on:
every Winograd layer of the network takes the two-dimensional form in the engine's operand form, operands that the form cannot
serve fall back to the row form, the workspace query covers whichever form serves the call, and the MFMA counts that bench.py
turns into `roofline.second` are the padded tile counts."""
import ctypes as C

import pytest

from gelslim_depth_amd import _lib as L

lib = L.lib
HS, WS, CS = [320, 160, 80, 40, 20], [427, 213, 106, 53, 26], [64, 128, 256, 512, 1024]


def src(c, h, w, *, pitch=None, slack=4, ptr=0x7F0000000000, off=(0, 0), affine=False):
    s = L.gsd_src()
    ws = pitch or w
    s.ptr = ptr
    s.scale = 0x7F1000000000 if affine else None
    s.shift = 0x7F1000100000 if affine else None
    s.C, s.H, s.W = c, h, w
    s.off_h, s.off_w = off
    s.relu = 1 if affine else 0
    s.w_stride = ws
    s.slack = slack
    s.c_stride = h * ws
    s.n_stride = c * h * ws
    return s


def form(segs, dy, ci, co, n, h, w):
    arr = L.src_array(segs)
    return lib.gsd_conv3x3_wgrad_form(arr, len(segs), C.byref(dy), ci, co, n, h, w)


def layers():
    out = [(0, 64, 0, 64)]
    for l in range(1, 5):
        out += [(l, CS[l - 1], 0, CS[l]), (l, CS[l], 0, CS[l])]
    for l in (3, 2, 1, 0):
        out += [(l, CS[l], CS[l], CS[l]), (l, CS[l], 0, CS[l])]
    return out


@pytest.mark.parametrize("n", [8, 32, 64])
def test_every_network_layer_takes_the_two_dimensional_form(n):
    for lvl, c0, c1, co in layers():
        h, w = HS[lvl], WS[lvl]
        segs = [src(c0, h, w, affine=True)]
        if c1:
            segs.append(src(c1, 2 * HS[lvl + 1], 2 * WS[lvl + 1], off=((h - 2 * HS[lvl + 1]) // 2, (w - 2 * WS[lvl + 1]) // 2)))
        dy = src(co, h, w, pitch=(w + 3) // 4 * 4, slack=0)
        assert form(segs, dy, c0 + c1, co, n, h, w) == 2, (lvl, c0, c1, co)
        need = lib.gsd_conv3x3_wgrad_workspace(n, h, w, c0 + c1, co)
        blocks = (co // (128 if co >= 128 else 64)) * ((c0 + c1) // (32 if co >= 128 else 64))
        splits = -(-256 // blocks)
        assert need >= splits * 9 * co * (c0 + c1)           # the 2-D form's slabs fit the queried workspace
        # executed MFMAs: k-steps x 24 frequencies per 16 x 16 channel pair, k-steps = padded tile groups of four
        cnt = lib.gsd_conv3x3_wgrad_mfma_count(2, n, h, w, c0 + c1, co)
        ty, tx = (h + 1) // 2, (w + 3) // 4
        ksteps = min(-(-ty // ky) * -(-tx // kx) for ky, kx in ((1, 4), (2, 2), (4, 1)))
        per = 24 * (co // 16) * ((c0 + c1) // 16)
        assert cnt % per == 0 and cnt // per >= n * ksteps
        assert cnt // per <= n * ksteps * 1.13               # the planner trades at most ~12 % more k-steps for cheaper ones
        assert cnt * 3 <= lib.gsd_conv3x3_wgrad_mfma_count(1, n, h, w, c0 + c1, co) * 2 * 1.2     # ~2/3 of the row form's


def test_operands_the_form_declines_go_to_the_row_form(monkeypatch):
    n, h, w, ci, co = 2, 40, 53, 64, 128
    good = [src(ci, h, w)]
    dy = src(co, h, w, pitch=56, slack=0)
    assert form(good, dy, ci, co, n, h, w) == 2
    assert form([src(ci, h, w, slack=0)], dy, ci, co, n, h, w) == 1                      # no slack around the activation
    assert form(good, src(co, h, w, slack=0), ci, co, n, h, w) == 1                      # dy rows not 16-byte aligned (W = 53)
    assert form(good, src(co, h, w, pitch=56, slack=0, ptr=0x7F0000000004), ci, co, n, h, w) == 1
    assert form([src(48, h, w)], dy, 48, co, n, h, w) == 1                               # Cin off the 32-channel block grid
    assert form(good, src(96, h, w, pitch=56, slack=0), ci, 96, n, h, w) == 1            # Cout neither 64 nor a multiple of 128
    assert form([src(32, h, w), src(32, h, w)], dy, ci, 64, n, h, w) == 1                # 64 x 64 blocks straddle the two segments
    assert form([src(64, h, w), src(64, h, w)], src(64, h, w, pitch=56, slack=0), 128, 64, n, h, w) == 2
    assert form([src(3, h, w)], dy, 3, co, n, h, w) == 0                                 # few channels: direct taps
    monkeypatch.setenv("GSD_WGRAD_W2D", "0")
    assert form(good, dy, ci, co, n, h, w) == 1
    monkeypatch.setenv("GSD_WGRAD_ALGO", "0")
    assert form(good, dy, ci, co, n, h, w) == 0
