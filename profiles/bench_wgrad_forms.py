"""fp32 conv3x3 dW on the U-Net's layer shapes (batch 32): run once with GSD_WGRAD_ALGO=0 (direct taps) and once with
GSD_WGRAD_ALGO=1 (Winograd F(4,3) rows); prints ms and algorithmic TFLOP/s per layer.
usage (GPU box): GSD_WGRAD_ALGO=1 PYTHONPATH=. python profiles/bench_wgrad_forms.py [batch]"""
import ctypes as C
import os
import sys
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = []
h, w = 320, 427
for lvl, c in enumerate([64, 128, 256, 512, 1024]):
    cin = 3 if lvl == 0 else c // 2
    shapes += [(lvl, cin, c, h, w), (lvl, c, c, h, w)]
    if lvl < 4:
        shapes.append((lvl, 2 * c, c, h, w))
    h, w = h // 2, w // 2
st = L.stream_ptr()
tot = 0.0
for lvl, ci, co, h, w in shapes:
    x = torch.randn(B, ci, h, w, device="cuda")
    dy = torch.randn(B, co, h, w, device="cuda")
    dw = torch.empty(co, ci, 3, 3, device="cuda")
    need = lib.gsd_conv3x3_wgrad_workspace(B, h, w, ci, co)
    ws = torch.empty(need, device="cuda")
    a_src, dy_src = L.src_array([L.make_src(x)]), L.make_src(dy)
    def run():
        check(lib.gsd_conv3x3_wgrad(a_src, 1, C.byref(dy_src), ci, co, dw.data_ptr(), ws.data_ptr(), need, B, h, w, st), "wgrad")
    run(); run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 4
    tot += ms
    gf = 2.0 * 9 * B * h * w * ci * co / 1e9
    print("L%d %4d->%4d %3dx%3d  %7.3f ms %6.1f TF  |dw| %.6e" % (lvl, ci, co, h, w, ms, gf / ms, dw.abs().sum().item()), flush=True)
print("algo=%s total %.2f ms" % (os.environ.get("GSD_WGRAD_ALGO"), tot))
