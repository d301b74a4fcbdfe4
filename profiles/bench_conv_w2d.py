"""fp32 conv3x3 on the U-Net's layer shapes: Winograd F(4,3) rows (gsd_conv3x3_w43) against the two-dimensional F(2x4,3x3)
(gsd_conv3x3_w2d), forward with a deferred-BatchNorm source + statistics, and the relative L1 of both against the direct-tap kernel.
usage (GPU box): python profiles/bench_conv_w2d.py [batch] [max level] [min level]"""
import sys
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
LMAX = int(sys.argv[2]) if len(sys.argv) > 2 else 4
LMIN = int(sys.argv[3]) if len(sys.argv) > 3 else 0
HS, WS = [320, 160, 80, 40, 20], [427, 213, 106, 53, 26]
C = [64, 128, 256, 512, 1024]
shapes = []
for lvl in range(LMIN, LMAX + 1):
    if lvl:
        shapes += [(lvl, C[lvl], C[lvl - 1]), (lvl, C[lvl - 1], C[lvl])]
    shapes += [(lvl, C[lvl], C[lvl])]
    if lvl < 4:
        shapes += [(lvl, C[lvl], 2 * C[lvl]), (lvl, 2 * C[lvl], C[lvl])]
st = L.stream_ptr()


def timed(fn, reps=6):
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


tot = [0.0, 0.0]
for lvl, m, k in shapes:
    h, w = HS[lvl], WS[lvl]
    x = L.slack_empty((B, k, h, w), "cuda")
    x.normal_()
    sc, sh = torch.rand(k, device="cuda") + 0.5, torch.randn(k, device="cuda") * 0.1
    wt = torch.randn(m, k, 3, 3, device="cuda") * (1.0 / (9 * k) ** 0.5)
    src = L.src_array([L.make_src(x, sc, sh, relu=True, slack=L.SLACK)])
    outs, ms = {}, {}
    for name, fn, mode, rows_fn in (("direct", lib.gsd_conv3x3, 0, lib.gsd_conv3x3_partial_rows),
                                     ("w43", lib.gsd_conv3x3_w43, 4, lib.gsd_conv3x3_w43_partial_rows),
                                     ("w2d", lib.gsd_conv3x3_w2d, 8, lib.gsd_conv3x3_w2d_partial_rows)):
        y = torch.full((B, m, h, w), float("nan"), device="cuda")
        rows = rows_fn(B, h, w, m)
        part = torch.zeros(rows * 2 * ((m + 127) // 128 * 128), device="cuda")
        wl = torch.empty(lib.gsd_weight_layout_size(mode, m, k), device="cuda")
        check(lib.gsd_weight_layout(mode, wt.data_ptr(), m, k, wl.data_ptr(), st), "layout")
        dst = L.dst_array([L.make_dst(y)])
        call = lambda: check(fn(src, 1, wl.data_ptr(), k, m, dst, 1, part.data_ptr(), B, h, w, st), name)
        if name == "direct":
            call()
        else:
            ms[name] = timed(call)
        torch.cuda.synchronize()
        outs[name] = y
    gf = 2.0 * 9 * B * h * w * k * m / 1e9
    ref = outs["direct"]
    e43 = ((outs["w43"] - ref).abs().sum() / ref.abs().sum()).item()
    e2d = ((outs["w2d"] - ref).abs().sum() / ref.abs().sum()).item()
    tot[0] += ms["w43"]
    tot[1] += ms["w2d"]
    print("L%d M%-4d K%-4d %3dx%-3d  w43 %7.3f ms %6.1f TF | w2d %7.3f ms %6.1f TF | x%.2f | relL1 vs direct: w43 %.1e  w2d %.1e  finite %s" % (
        lvl, m, k, h, w, ms["w43"], gf / ms["w43"], ms["w2d"], gf / ms["w2d"], ms["w43"] / ms["w2d"], e43, e2d,
        bool(torch.isfinite(outs["w2d"]).all())), flush=True)
print("total w43 %.2f ms, w2d %.2f ms" % tuple(tot))
