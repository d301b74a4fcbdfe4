"""How conv3x3_w43 / conv3x3_w2d launches scale with the number of blocks (one to a few per CU): 160x213, M = 128, K = 128, batch 1..16.
usage (GPU box): python profiles/bench_conv_grid.py"""
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
st = L.stream_ptr()
h, w, m, k = 160, 213, 128, 128


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for B in (1, 2, 3, 4, 6, 8, 12, 16):
    x = L.slack_empty((B, k, h, w), "cuda")
    x.normal_()
    wt = torch.randn(m, k, 3, 3, device="cuda") * 0.03
    src = L.src_array([L.make_src(x, slack=L.SLACK)])
    y = torch.empty((B, m, h, w), device="cuda")
    dst = L.dst_array([L.make_dst(y)])
    out = []
    for name, fn, mode in (("w43", lib.gsd_conv3x3_w43, 4), ("w2d", lib.gsd_conv3x3_w2d, 8)):
        wl = torch.empty(lib.gsd_weight_layout_size(mode, m, k), device="cuda")
        check(lib.gsd_weight_layout(mode, wt.data_ptr(), m, k, wl.data_ptr(), st), "layout")
        out.append(timed(lambda: check(fn(src, 1, wl.data_ptr(), k, m, dst, 1, None, B, h, w, st), name)))
    print("batch %2d  blocks %5d  w43 %.3f ms  w2d %.3f ms" % (B, B * 140 * 2, out[0], out[1]), flush=True)
