"""Where a wave of conv3x3_w43_kernel spends its cycles (diagnostic build with s_memtime stamps, profiles/build_diag.sh).
usage (GPU box): GSD_LIB_PATH=profiles/ubench/libgsd_diag.so PYTHONPATH=. python profiles/stamp_conv.py [ci co h w [B]]"""
import ctypes as C
import os
import sys
import numpy as np
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
args = [int(a) for a in sys.argv[1:]]
shapes = [tuple(args[:4])] if len(args) >= 4 else [(64, 64, 320, 427), (128, 128, 160, 213), (256, 256, 80, 106), (512, 512, 40, 53), (1024, 1024, 20, 26)]
B = args[4] if len(args) > 4 else 32
st = L.stream_ptr()
NAMES = ["barrier+vmcnt wait", "first reads+transform", "5 k-steps with DMA issue (20 MFMA)", "13 k-steps (52 MFMA)", "epilogue", "prologue", "DMA slots alone (W43_STAMP_DMA builds)"]
raw = C.CDLL(L.LIB_PATH)
raw.gsd_w43_set_stamp_buffer.argtypes = [C.c_void_p]
for ci, co, h, w in shapes:
    x = torch.randn(B, ci, h, w, device="cuda")
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    y = torch.empty(B, co, h, w, device="cuda")
    src, dst = L.src_array([L.make_src(x, sc, sh, relu=True)]), L.dst_array([L.make_dst(y)])
    wl = torch.empty(lib.gsd_weight_layout_size(4, co, ci), device="cuda")
    check(lib.gsd_weight_layout(4, wt.data_ptr(), co, ci, wl.data_ptr(), st), "layout")
    nblocks = lib.gsd_conv3x3_w43_partial_rows(B, h, w, co) // 4 * ((co + 63) // 64)
    buf = torch.zeros(nblocks * 4 * 8 + 64, dtype=torch.int64, device="cuda")
    for x4 in ("0", "1"):
        os.environ["GSD_W43_X4"] = x4
        raw.gsd_w43_set_stamp_buffer(None)
        for _ in range(2):
            check(lib.gsd_conv3x3_w43(src, 1, wl.data_ptr(), ci, co, dst, 1, None, B, h, w, st), "conv")
        buf.zero_()
        raw.gsd_w43_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.gsd_conv3x3_w43(src, 1, wl.data_ptr(), ci, co, dst, 1, None, B, h, w, st), "conv")
        e1.record()
        torch.cuda.synchronize()
        raw.gsd_w43_set_stamp_buffer(None)
        a = buf[:nblocks * 4 * 8].cpu().numpy().reshape(-1, 8).astype(np.float64)
        a = a[a.sum(axis=1) > 0]
        tot = a[:, :7].sum(axis=1)
        nch = (ci + 3) // 4
        print("%4d->%4d %3dx%3d B%d x4=%s  %.3f ms (stamped build), %d waves, %.0f cycles/wave, chunks %d" % (ci, co, h, w, B, x4, e0.elapsed_time(e1), len(a), tot.mean(), nch))
        for i, nm in enumerate(NAMES):
            per = a[:, i].mean() / (nch if (i < 4 or i == 6) else 1)
            print("    %-36s %5.1f %%   %8.0f cycles per %s" % (nm, 100 * a[:, i].sum() / tot.sum(), per, "chunk" if (i < 4 or i == 6) else "wave"))
