"""fp32 Winograd conv3x3 forward on the U-Net's layer shapes: dword halo gathers vs aligned 16-byte halo pieces (GSD_W43_X4),
plus a correctness check of both against the direct-tap kernel.
usage (GPU box): python profiles/bench_conv_x4.py [batch] [W0]     (W0 % 32 == 0 keeps every level's rows 16-byte aligned)"""
import os
import sys
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H0 = 320
W0 = int(sys.argv[2]) if len(sys.argv) > 2 else 448
shapes = []
h, w = H0, W0
for lvl, c in enumerate([64, 128, 256, 512, 1024]):
    cin = 3 if lvl == 0 else c // 2
    if lvl:
        shapes.append((lvl, cin, c, h, w))
    shapes.append((lvl, c, c, h, w))
    if lvl < 4:
        shapes.append((lvl, 2 * c, c, h, w))      # decoder conv1 on the concat
    h, w = h // 2, w // 2
st = L.stream_ptr()
tot = [0.0, 0.0]
for lvl, ci, co, h, w in shapes:
    x = torch.randn(B, ci, h, w, device="cuda")
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    y = torch.empty(B, co, h, w, device="cuda")
    src, dst = L.src_array([L.make_src(x, sc, sh, relu=True)]), L.dst_array([L.make_dst(y)])
    wl = torch.empty(lib.gsd_weight_layout_size(4, co, ci), device="cuda")
    check(lib.gsd_weight_layout(4, wt.data_ptr(), co, ci, wl.data_ptr(), st), "layout")
    wl0 = torch.empty(lib.gsd_weight_layout_size(0, co, ci), device="cuda")
    check(lib.gsd_weight_layout(0, wt.data_ptr(), co, ci, wl0.data_ptr(), st), "layout")
    check(lib.gsd_conv3x3(src, 1, wl0.data_ptr(), ci, co, dst, 1, None, B, h, w, st), "conv")
    ref = y.clone()
    res, errs = [], []
    for x4 in ("0", "1"):
        os.environ["GSD_W43_X4"] = x4
        y.zero_()
        for _ in range(2):
            check(lib.gsd_conv3x3_w43(src, 1, wl.data_ptr(), ci, co, dst, 1, None, B, h, w, st), "conv")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            check(lib.gsd_conv3x3_w43(src, 1, wl.data_ptr(), ci, co, dst, 1, None, B, h, w, st), "conv")
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
        errs.append(((y - ref).abs().sum() / ref.abs().sum()).item())
    gf = 2.0 * 9 * B * h * w * ci * co / 1e9
    tot[0] += res[0]; tot[1] += res[1]
    print("L%d %4d->%4d %3dx%3d  dword %7.3f ms %6.1f TF | x4 %7.3f ms %6.1f TF | x%.3f  relL1 %.1e %.1e" % (
        lvl, ci, co, h, w, res[0], gf / res[0], res[1], gf / res[1], res[0] / res[1], errs[0], errs[1]), flush=True)
print("total dword %.2f ms, x4 %.2f ms" % tuple(tot))
