#!/usr/bin/env python3
"""Per-shape timing of the bf16 weight-gradient kernel (gsd_bf16_wgrad, 3x3) at the batch-32 layer shapes."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gelslim_depth_amd import _lib as L  # noqa: E402

N = int(os.environ.get("N", "32"))
SHAPES = [(320, 427, 64, 64), (320, 427, 128, 64), (160, 213, 64, 128), (160, 213, 128, 128), (160, 213, 256, 128),
          (80, 106, 128, 256), (80, 106, 256, 256), (80, 106, 512, 256), (40, 53, 256, 512), (40, 53, 512, 512),
          (40, 53, 1024, 512), (20, 26, 512, 1024), (20, 26, 1024, 1024)]
T3Y = L.int_array([t // 3 - 1 for t in range(9)])
T3X = L.int_array([t % 3 - 1 for t in range(9)])
tot_f = tot_t = 0.0
for h, w, k, m in SHAPES:
    a = torch.randn((N, h, w, k), device="cuda").to(torch.bfloat16)
    dy = torch.randn((N, h, w, m), device="cuda").to(torch.bfloat16)
    nws = L.lib.gsd_bf16_wgrad_workspace(9, N, h, w, m, k)
    ws = torch.empty((nws,), device="cuda")
    dw = torch.empty((m * k * 9,), device="cuda")
    da, db = L.make_nhwc(dy), L.make_nhwc(a)

    def run():
        L.check(L.lib.gsd_bf16_wgrad(C.byref(da), C.byref(db), 9, 1, T3Y, T3X, dw.data_ptr(), k, ws.data_ptr(), nws, L.stream_ptr()), "wgrad")
    for _ in range(2):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * m * k * 9 * N * h * w
    tot_f += fl
    tot_t += ms
    print(f"{h}x{w} K{k} M{m}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  (slab {nws * 4 / 1e6:.0f} MB)", flush=True)
print(f"total {tot_t:.2f} ms, {tot_f / tot_t / 1e9:.1f} TFLOP/s")
