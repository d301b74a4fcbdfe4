"""A/B of tuning variants of the fp32 Winograd conv3x3 kernel on the U-Net's layer shapes, all in ONE process (interleaved
rounds), each checked against the direct-tap kernel.
usage (GPU box): PYTHONPATH=. python profiles/bench_conv_ab.py B W0 "NAME:ENV=V,ENV=V;NAME:..." [rounds]
   e.g. python profiles/bench_conv_ab.py 32 427 "base:GSD_W43_NL=0;lw1:GSD_W43_NL=1;lw2:GSD_W43_NL=2" """
import os
import sys
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
B = int(sys.argv[1])
W0 = int(sys.argv[2])
variants = []
for spec in sys.argv[3].split(";"):
    name, _, envs = spec.partition(":")
    variants.append((name, dict(e.split("=") for e in envs.split(",") if e)))
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
KEYS = sorted({k for _, e in variants for k in e})
H0 = 320
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
tot = {n: 0.0 for n, _ in variants}
for lvl, ci, co, h, w in shapes:
    x = L.slack_empty((B, ci, h, w), "cuda")   # 4 readable floats either side: lets the kernel move the halo as 16-byte pieces
    x.copy_(torch.randn(B, ci, h, w, device="cuda"))
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    y = torch.empty(B, co, h, w, device="cuda")
    src, dst = L.src_array([L.make_src(x, sc, sh, relu=True, slack=L.SLACK)]), L.dst_array([L.make_dst(y)])
    wl = torch.empty(lib.gsd_weight_layout_size(4, co, ci), device="cuda")
    check(lib.gsd_weight_layout(4, wt.data_ptr(), co, ci, wl.data_ptr(), st), "layout")
    wl0 = torch.empty(lib.gsd_weight_layout_size(0, co, ci), device="cuda")
    check(lib.gsd_weight_layout(0, wt.data_ptr(), co, ci, wl0.data_ptr(), st), "layout")
    check(lib.gsd_conv3x3(src, 1, wl0.data_ptr(), ci, co, dst, 1, None, B, h, w, st), "conv")
    ref = y.clone()
    best = {n: 1e9 for n, _ in variants}
    errs = {}
    for rnd in range(rounds):
        for name, env in variants:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            if rnd == 0:
                y.zero_()
            check(lib.gsd_conv3x3_w43(src, 1, wl.data_ptr(), ci, co, dst, 1, None, B, h, w, st), "conv")
            torch.cuda.synchronize()
            if rnd == 0:
                errs[name] = ((y - ref).abs().sum() / ref.abs().sum()).item()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                check(lib.gsd_conv3x3_w43(src, 1, wl.data_ptr(), ci, co, dst, 1, None, B, h, w, st), "conv")
            e1.record()
            torch.cuda.synchronize()
            best[name] = min(best[name], e0.elapsed_time(e1) / 4)
    gf = 2.0 * 9 * B * h * w * ci * co / 1e9
    line = "L%d %4d->%4d %3dx%3d " % (lvl, ci, co, h, w)
    for name, _ in variants:
        tot[name] += best[name]
        line += "| %s %6.3f ms %5.1f TF err %.0e " % (name, best[name], gf / best[name], errs[name])
    print(line, flush=True)
print("TOTAL " + "  ".join("%s %.2f ms" % (n, tot[n]) for n, _ in variants))
