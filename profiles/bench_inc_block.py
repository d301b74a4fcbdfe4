#!/usr/bin/env python3
"""The bf16 `inc` double-conv forward (3 -> 64 -> 64 @320x427, batch N) kernel by kernel, fused (gsd_bf16_inc.hip) and unfused, each
launch timed with HIP events on its own; with a -DINC_STAMP=1 diagnostic build (profiles/build_diag_one.sh gsd_bf16_inc.hip
"-DINC_STAMP=1" incstamp; GSD_LIB_PATH=profiles/ubench/libgsd_incstamp.so GSD_DIAG_STAMPS=1) also where a wave of the fused kernel
spends its cycles."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gelslim_depth_amd import _lib as L  # noqa: E402

N, H, W, M, CIN = int(os.environ.get("N", "32")), 320, 427, 64, 3
lib = L.lib
x = torch.rand((N, CIN, H, W), device="cuda")
w0 = torch.randn((M, CIN, 3, 3), device="cuda") * 0.3
w1 = torch.randn((M, M, 3, 3), device="cuda") * 0.05
img0 = torch.zeros(lib.gsd_bf16_weight_image_size(2, M, CIN), dtype=torch.bfloat16, device="cuda")
img1 = torch.zeros(lib.gsd_bf16_weight_image_size(0, M, M), dtype=torch.bfloat16, device="cuda")
L.check(lib.gsd_bf16_weight_image(2, w0.data_ptr(), M, CIN, img0.data_ptr(), L.stream_ptr()), "w0")
L.check(lib.gsd_bf16_weight_image(0, w1.data_ptr(), M, M, img1.data_ptr(), L.stream_ptr()), "w1")
mp = lib.gsd_bf16_conv_mpad(M)
y0 = torch.empty((N, H, W, M), dtype=torch.bfloat16, device="cuda")
a0 = torch.empty_like(y0)
y1 = torch.empty_like(y0)
cat = torch.empty((N, H, W, 2 * M), dtype=torch.bfloat16, device="cuda")
pooled = torch.empty((N, H // 2, W // 2, M), dtype=torch.bfloat16, device="cuda")
part = torch.empty((4096 * 2 * mp,), device="cuda")
scale, shift = torch.rand(M, device="cuda") + 0.5, torch.randn(M, device="cuda") * 0.1
mean, invstd = torch.randn(M, device="cuda") * 0.1, torch.rand(M, device="cuda") + 0.5
c1, c2 = torch.randn(M, device="cuda") * 0.01, torch.randn(M, device="cuda") * 0.01
dy0, da0, dy1, dcat, dpool = L.make_nhwc(y0), L.make_nhwc(a0), L.make_nhwc(y1), L.make_nhwc(cat, 0, M), L.make_nhwc(pooled)
need = lib.gsd_bf16_wgrad_first_workspace(N, H, W, M)
ws = torch.empty(need, device="cuda")
dw = torch.empty((M, CIN, 3, 3), device="cuda")
st = L.stream_ptr()
ops = {
    "first conv, storing y0": lambda: lib.gsd_bf16_conv3x3_first(x.data_ptr(), N, CIN, H, W, img0.data_ptr(), C.byref(dy0), M, part.data_ptr(), None, None, st),
    "first conv, statistics only": lambda: lib.gsd_bf16_conv3x3_first(x.data_ptr(), N, CIN, H, W, img0.data_ptr(), None, M, part.data_ptr(), None, None, st),
    "bn_apply y0 -> a0": lambda: lib.gsd_bf16_bn_apply(C.byref(dy0), scale.data_ptr(), shift.data_ptr(), C.byref(da0), 1, st),
    "conv3x3 64->64 (DMA kernel)": lambda: lib.gsd_bf16_conv3x3(C.byref(da0), img1.data_ptr(), C.byref(dy1), M, M, part.data_ptr(), None, st),
    "inc_conv (fused)": lambda: lib.gsd_bf16_inc_conv(x.data_ptr(), N, CIN, H, W, img0.data_ptr(), scale.data_ptr(), shift.data_ptr(), img1.data_ptr(), C.byref(da0), C.byref(dy1), part.data_ptr(), st),
    "bn_apply y1 -> cat": lambda: lib.gsd_bf16_bn_apply(C.byref(dy1), scale.data_ptr(), shift.data_ptr(), C.byref(dcat), 1, st),
    "bn_apply_pool y1 -> cat, pooled": lambda: lib.gsd_bf16_bn_apply_pool(C.byref(dy1), scale.data_ptr(), shift.data_ptr(), C.byref(dcat), C.byref(dpool), st),
    "maxpool2 cat -> pooled": lambda: lib.gsd_bf16_maxpool2(C.byref(dcat), C.byref(dpool), st),
    "first_bn_bwd_reduce (y0 recomputed)": lambda: lib.gsd_bf16_first_bn_bwd_reduce(x.data_ptr(), N, CIN, H, W, img0.data_ptr(), C.byref(da0), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), part.data_ptr(), st),
    "wgrad_first (dz, y0)": lambda: lib.gsd_bf16_wgrad_first(x.data_ptr(), N, CIN, H, W, C.byref(da0), C.byref(dy0), scale.data_ptr(), mean.data_ptr(), invstd.data_ptr(), c1.data_ptr(), c2.data_ptr(), dw.data_ptr(), ws.data_ptr(), need, st),
    "wgrad_first_recompute (da)": lambda: lib.gsd_bf16_wgrad_first_recompute(x.data_ptr(), N, CIN, H, W, img0.data_ptr(), C.byref(da0), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), c1.data_ptr(), c2.data_ptr(), dw.data_ptr(), ws.data_ptr(), need, st),
}
only = os.environ.get("ONLY")
for name, fn in ops.items():
    if only and only not in name:
        continue
    for _ in range(2):
        L.check(fn(), name)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        L.check(fn(), name)
    b.record()
    torch.cuda.synchronize()
    print(f"{name:40s} {a.elapsed_time(b) / 5:.4f} ms", flush=True)
    if name.startswith("inc_conv") and os.environ.get("GSD_DIAG_STAMPS"):
        import numpy as np
        buf = (C.c_ulonglong * (8 * 256))()
        lib.gsd_diag_inc_stamps(buf, 256)
        raw = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)
        w2, w3 = (raw[:, 7] & np.uint64(0xffffffff)).astype(np.float64), (raw[:, 7] >> np.uint64(32)).astype(np.float64)
        t = raw.astype(np.float64)
        keep = t[:, 4] > 0
        t, w2, w3 = t[keep], w2[keep], w3[keep]
        it = t[:, 4]
        print(f"   per item (median over {len(t)} blocks, {np.median(it):.0f} items per block): rebuild {np.median(t[:, 0] / it):.0f}  K loop "
              f"{np.median(t[:, 1] / it):.0f}  put_x + epilogue {np.median(t[:, 2] / it):.0f}  barrier waits after them {np.median(t[:, 3] / it):.0f} / {np.median(w2 / it):.0f} / {np.median(w3 / it):.0f} cycles; "
              f"block life {np.median(t[:, 5]) / 1e3:.0f} kcycles at {np.median(t[:, 5] / t[:, 6]) * 0.1:.2f} GHz")
