#!/usr/bin/env python3
"""Per-shape timing of the bf16 conv3x3 kernel (gsd_bf16_conv3x3) at the batch-32 layer shapes."""
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
tot_f = tot_t = 0.0
for h, w, k, m in SHAPES:
    x = torch.randn((N, h, w, k), device="cuda").to(torch.bfloat16)
    mp = L.lib.gsd_bf16_conv_mpad(m)
    wt = (torch.randn((9, mp, k), device="cuda") / (3 * k ** 0.5)).to(torch.bfloat16)
    out = torch.empty((N, h, w, m), device="cuda", dtype=torch.bfloat16)
    rows = L.lib.gsd_bf16_conv_partial_rows(N, h, w, m)
    part = torch.empty((rows, 2 * mp), device="cuda")
    din, dout = L.make_nhwc(x), L.make_nhwc(out)

    def run():
        L.check(L.lib.gsd_bf16_conv3x3(C.byref(din), wt.data_ptr(), C.byref(dout), k, m, None if os.environ.get("NOSTATS") else part.data_ptr(), None, L.stream_ptr()), "conv")
    for _ in range(2):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    fl = 2.0 * m * k * 9 * N * h * w
    tot_f += fl
    tot_t += ms
    clk = ""
    if os.environ.get("GSD_DIAG_STAMPS"):   # a -DGCONV_STAMP=1 build (profiles/build_diag_one.sh): the clock the chip held in the kernel
        import numpy as np
        buf = (C.c_ulonglong * (2 * 256))()
        L.lib.gsd_diag_gconv_stamps(buf, 256)
        a_ = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
        a_ = a_[a_[:, 1] > 0]
        clk = f"  in-kernel clock {np.median(a_[:, 0] / a_[:, 1]) * 0.1:.2f} GHz over {len(a_)} blocks, {np.median(a_[:, 0]) / 1e3:.1f} kcycles per block (max {a_[:, 0].max() / 1e3:.1f})"
    print(f"{h}x{w} K{k} M{m}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s{clk}", flush=True)
print(f"total {tot_t:.2f} ms, {tot_f / tot_t / 1e9:.1f} TFLOP/s")
