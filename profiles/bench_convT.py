"""fp32 ConvTranspose2d(k2,s2) forward / dX / dW on the U-Net's four upsamplers (batch 32): ms and TFLOP/s per launch.
usage (GPU box): PYTHONPATH=. python profiles/bench_convT.py [batch]"""
import ctypes as C
import sys
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
st = L.stream_ptr()
tot = [0.0, 0.0, 0.0]
for ci, h, w in ((1024, 20, 26), (512, 40, 53), (256, 80, 106), (128, 160, 213)):
    co = ci // 2
    x = torch.randn(B, ci, h, w, device="cuda")
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
    wt = torch.randn(ci, co, 2, 2, device="cuda") * 0.05
    bias = torch.randn(co, device="cuda")
    y = torch.empty(B, co, 2 * h, 2 * w, device="cuda")
    dy = L.slack_empty((B, co, 2 * h, 2 * w), "cuda")
    dy.normal_()
    dx = torch.empty(B, ci, h, w, device="cuda")
    dw, db = torch.empty(ci, co, 2, 2, device="cuda"), torch.empty(co, device="cuda")
    wf = torch.empty(lib.gsd_weight_layout_size(6, co, ci), device="cuda")
    sdy = L.make_src(dy, slack=L.SLACK)
    mode_d = lib.gsd_convT2x2_dgrad_layout(C.byref(sdy), ci, co, B, h, w)      # 7: LDS-DMA kernel (GSD_CONVT_DG_DMA=0: 3)
    wd = torch.empty(lib.gsd_weight_layout_size(mode_d, co, ci), device="cuda")
    check(lib.gsd_weight_layout(6, wt.data_ptr(), co, ci, wf.data_ptr(), st), "layout")
    check(lib.gsd_weight_layout(mode_d, wt.data_ptr(), co, ci, wd.data_ptr(), st), "layout")
    need = lib.gsd_convT2x2_wgrad_workspace(B, h, w, ci, co)
    ws = torch.empty(need, device="cuda")
    s, d = L.make_src(x, sc, sh, relu=True), L.make_dst(y)
    ddx = L.make_dst(dx)
    fns = [lambda: check(lib.gsd_convT2x2(C.byref(s), wf.data_ptr(), bias.data_ptr(), ci, co, C.byref(d), B, h, w, st), "fwd"),
           lambda: check(lib.gsd_convT2x2_dgrad(C.byref(sdy), wd.data_ptr(), ci, co, C.byref(ddx), B, h, w, st), "dgrad"),
           lambda: check(lib.gsd_convT2x2_wgrad(C.byref(s), C.byref(sdy), ci, co, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), need, B, h, w, st), "wgrad")]
    gf = 2.0 * 4 * ci * co * B * h * w / 1e9
    line = "%4d->%4d %3dx%3d " % (ci, co, h, w)
    for i, fn in enumerate(fns):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tot[i] += ms
        line += "| %s %6.3f ms %5.1f TF " % (("fwd", "dX ", "dW ")[i], ms, gf / ms)
    print(line, flush=True)
print("TOTAL fwd %.2f  dX %.2f  dW %.2f ms" % tuple(tot))
