"""Where a wave of wgrad3x3_w2d_kernel spends its cycles (diagnostic build:
bash profiles/build_diag_one.sh gsd_wgrad_w2d.hip "-DGSD_WG2D_STAMPS" wg2d_stamps).
usage (GPU box): GSD_LIB_PATH=$PWD/profiles/ubench/libgsd_wg2d_stamps.so PYTHONPATH=. python profiles/stamp_wgrad_w2d.py [ci co h w [B]]"""
import ctypes as C
import sys
import numpy as np
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
args = [int(a) for a in sys.argv[1:]]
shapes = [tuple(args[:4])] if len(args) >= 4 else [(64, 64, 320, 427), (128, 128, 160, 213), (256, 256, 80, 106), (512, 512, 40, 53), (1024, 1024, 20, 26)]
B = args[4] if len(args) > 4 else 32
st = L.stream_ptr()
NAMES = ["raw reads + fill issue", "transform (+ its LDS writes landed)", "operand reads + 48 MFMAs", "vmcnt(0): fills landed", "barrier", "prologue + epilogue + loop control"]
raw = C.CDLL(L.LIB_PATH)
raw.gsd_wg2d_set_stamp_buffer.argtypes = [C.c_void_p]
for ci, co, h, w in shapes:
    x = L.slack_empty((B, ci, h, w), "cuda")
    x.normal_()
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
    p = (w + 3) // 4 * 4
    dyb = torch.zeros((B, co, h, p), device="cuda")
    dyb[..., :w].normal_()
    dy = dyb[..., :w]
    dw = torch.empty(co, ci, 3, 3, device="cuda")
    need = lib.gsd_conv3x3_wgrad_workspace(B, h, w, ci, co)
    ws = torch.empty(need, device="cuda")
    a_src, dy_src = L.src_array([L.make_src(x, sc, sh, relu=True, slack=L.SLACK)]), L.make_src(dy)
    buf = torch.zeros(4096 * 8 * 6, dtype=torch.int64, device="cuda")

    def run():
        check(lib.gsd_conv3x3_wgrad(a_src, 1, C.byref(dy_src), ci, co, dw.data_ptr(), ws.data_ptr(), need, B, h, w, st), "wgrad")
    raw.gsd_wg2d_set_stamp_buffer(None)
    run(); run()
    buf.zero_()
    raw.gsd_wg2d_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    raw.gsd_wg2d_set_stamp_buffer(None)
    a = buf.cpu().numpy().reshape(-1, 6).astype(np.float64)
    a = a[a.sum(axis=1) > 0]
    tot = a.sum(axis=1)
    print("%4d->%4d %3dx%3d B%d  %.3f ms incl. slab reduction (stamped build), %d waves, %.0f ticks/wave" % (ci, co, h, w, B, e0.elapsed_time(e1), len(a), tot.mean()))
    for i, nm in enumerate(NAMES):
        print("    %-40s %5.1f %%" % (nm, 100 * a[:, i].sum() / tot.sum()))
