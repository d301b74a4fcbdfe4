"""fp32 Winograd conv3x3 dW on the U-Net's layer shapes (batch 32) with dy from a row-pitched buffer (what the engine feeds it).
usage (GPU box): PYTHONPATH=. python profiles/bench_wgrad_pitched.py [batch]"""
import ctypes as C
import sys
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = []
h, w = 320, 427
for lvl, c in enumerate([64, 128, 256, 512, 1024]):
    cin = 3 if lvl == 0 else c // 2
    if lvl:
        shapes.append((lvl, cin, c, h, w))
    shapes.append((lvl, c, c, h, w))
    if lvl < 4:
        shapes.append((lvl, 2 * c, c, h, w))
    h, w = h // 2, w // 2
st = L.stream_ptr()
tot = 0.0
for lvl, ci, co, h, w in shapes:
    x = L.slack_empty((B, ci, h, w), "cuda")   # 4 readable floats either side: the dW kernel may move windows as 16-byte pieces
    x.copy_(torch.randn(B, ci, h, w, device="cuda"))
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
    dy = L.pitched_empty((B, co, h, w), "cuda")
    dy.copy_(torch.randn(B, co, h, w, device="cuda"))
    dw = torch.empty(co, ci, 3, 3, device="cuda")
    need = lib.gsd_conv3x3_wgrad_workspace(B, h, w, ci, co)
    ws = torch.empty(need, device="cuda")
    a_src, dy_src = L.src_array([L.make_src(x, sc, sh, relu=True, slack=L.SLACK)]), L.make_src(dy)

    def run():
        check(lib.gsd_conv3x3_wgrad(a_src, 1, C.byref(dy_src), ci, co, dw.data_ptr(), ws.data_ptr(), need, B, h, w, st), "wgrad")
    run(); run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4)
    tot += best
    gf = 2.0 * 9 * B * h * w * ci * co / 1e9
    print("L%d %4d->%4d %3dx%3d  %7.3f ms %6.1f TF  |dw| %.6e" % (lvl, ci, co, h, w, best, gf / best, dw.abs().sum().item()), flush=True)
print("total %.2f ms" % tot)
