"""fp32 conv3x3 forward on the U-Net's layer shapes (batch 32, 320x427 at level 0): direct taps vs Winograd F(4,3) rows.
usage (GPU box): python profiles/bench_conv_forms.py [batch]   -> one line per layer: ms and algorithmic TFLOP/s of both forms"""
import sys
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H0, W0 = 320, 427
shapes = []
h, w = H0, W0
for lvl, c in enumerate([64, 128, 256, 512, 1024]):
    cin = 3 if lvl == 0 else c // 2
    shapes += [(lvl, cin, c, h, w), (lvl, c, c, h, w)]
    if lvl < 4:
        shapes.append((lvl, 2 * c, c, h, w))      # decoder conv1 on the concat
    h, w = h // 2, w // 2
st = L.stream_ptr()
tot = [0.0, 0.0]
for lvl, ci, co, h, w in shapes:
    x = torch.randn(B, ci, h, w, device="cuda")
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    y = torch.empty(B, co, h, w, device="cuda")
    src, dst = L.src_array([L.make_src(x)]), L.dst_array([L.make_dst(y)])
    res = []
    outs = []
    for algo in (0, 1):
        fn = lib.gsd_conv3x3_w43 if algo else lib.gsd_conv3x3
        mode = 4 if algo else 0
        wl = torch.empty(lib.gsd_weight_layout_size(mode, co, ci), device="cuda")
        check(lib.gsd_weight_layout(mode, wt.data_ptr(), co, ci, wl.data_ptr(), st), "layout")
        for _ in range(2):
            check(fn(src, 1, wl.data_ptr(), ci, co, dst, 1, None, B, h, w, st), "conv")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            check(fn(src, 1, wl.data_ptr(), ci, co, dst, 1, None, B, h, w, st), "conv")
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
        outs.append(y.clone())
    gf = 2.0 * 9 * B * h * w * ci * co / 1e9
    err = ((outs[0] - outs[1]).abs().sum() / outs[0].abs().sum()).item()
    tot[0] += res[0]; tot[1] += res[1]
    print("L%d %4d->%4d %3dx%3d  direct %7.3f ms %6.1f TF | w43 %7.3f ms %6.1f TF | x%.2f  algo=%d  relL1 %.1e" % (
        lvl, ci, co, h, w, res[0], gf / res[0], res[1], gf / res[1], res[0] / res[1],
        lib.gsd_conv3x3_algo(B, h, w, ci, co), err), flush=True)
print("total direct %.2f ms, w43 %.2f ms" % tuple(tot))
