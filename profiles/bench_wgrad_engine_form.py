"""fp32 conv3x3 dW on the U-Net's 17 Winograd layers in the form the ENGINE launches them: activation segments with deferred
BatchNorm + ReLU and slack (16-byte window pieces), the two-segment decoder form, dy from the row-pitched buffer (aligned 16-byte
pieces).  One line per layer: ms and algorithmic TFLOP/s.  GSD_WG43_TRACE=1 adds the planner's choice per launch (stderr).
usage (GPU box): python profiles/bench_wgrad_engine_form.py [batch [layer,layer,...]]"""
import ctypes as C
import sys
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
HS, WS = [320, 160, 80, 40, 20], [427, 213, 106, 53, 26]
Cs = [64, 128, 256, 512, 1024]
layers = [("inc.c1", 0, 64, 0, 64, False)]
for l in range(1, 5):
    layers += [(f"down{l-1}.c0", l, Cs[l - 1], 0, Cs[l], True), (f"down{l-1}.c1", l, Cs[l], 0, Cs[l], False)]
for j, l in enumerate((3, 2, 1, 0)):
    layers += [(f"up{j}.c0", l, Cs[l], Cs[l], Cs[l], False), (f"up{j}.c1", l, Cs[l], 0, Cs[l], False)]
if len(sys.argv) > 2:
    layers = [l for l in layers if l[0] in sys.argv[2].split(",")]
st = L.stream_ptr()
tot = 0.0
for name, lvl, c0, c1, co, pooled in layers:
    h, w = HS[lvl], WS[lvl]
    ci = c0 + c1
    x0 = L.slack_empty((B, c0, h, w), "cuda")
    x0.normal_()
    keep = [x0]
    if pooled:
        segs = [L.make_src(x0, slack=L.SLACK)]
    else:
        sc, sh = torch.rand(c0, device="cuda") + 0.5, torch.randn(c0, device="cuda") * 0.1
        keep += [sc, sh]
        segs = [L.make_src(x0, sc, sh, relu=True, slack=L.SLACK)]
    if c1:
        uh, uw = 2 * HS[lvl + 1], 2 * WS[lvl + 1]
        up = L.slack_empty((B, c1, uh, uw), "cuda")
        up.normal_()
        keep.append(up)
        segs.append(L.make_src(up, off=((h - uh) // 2, (w - uw) // 2), slack=L.SLACK))
    p = (w + 3) // 4 * 4
    dyb = torch.zeros((B, co, h, p), device="cuda")
    dyb[..., :w].normal_()
    dy = dyb[..., :w]
    dw = torch.empty(co, ci, 3, 3, device="cuda")
    need = lib.gsd_conv3x3_wgrad_workspace(B, h, w, ci, co)
    ws = torch.empty(need, device="cuda")
    a_src, dy_src = L.src_array(segs), L.make_src(dy)

    def run():
        check(lib.gsd_conv3x3_wgrad(a_src, len(segs), C.byref(dy_src), ci, co, dw.data_ptr(), ws.data_ptr(), need, B, h, w, st), "wgrad")
    run(); run(); run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tot += ms
    gf = 2.0 * 9 * B * h * w * ci * co / 1e9
    print("%-9s L%d %4d->%4d %3dx%3d  %7.3f ms %6.1f TF" % (name, lvl, ci, co, h, w, ms, gf / ms), flush=True)
print("total %.2f ms" % tot)
