#!/usr/bin/env python3
"""The fp32 first convolution (3 -> 64 @320x427, batch N): the general direct-tap kernel against gsd_conv3x3_first, a store-stream
kernel that was built in round 4 and NOT kept (profiles/ubench/gsd_conv3x3_first_store_stream.hip: add it to build.py's SOURCES,
declare it in include/gsd.h and _lib.py to run this).  Result of the round: 0.5388 vs 0.5384 ms."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gelslim_depth_amd import _lib as L  # noqa: E402

N, H, W, M, CIN = int(os.environ.get("N", "32")), 320, 427, 64, 3
lib = L.lib
x = torch.rand((N, CIN, H, W), device="cuda")
w = torch.randn((M, CIN, 3, 3), device="cuda") * 0.3
out = L.slack_empty((N, M, H, W), "cuda")
wt = torch.zeros(lib.gsd_weight_layout_size(0, M, CIN), device="cuda")
L.check(lib.gsd_weight_layout(0, w.data_ptr(), M, CIN, wt.data_ptr(), L.stream_ptr()), "layout")
rows = max(lib.gsd_conv3x3_partial_rows(N, H, W, M), 2048)
part = torch.empty((rows * 2 * 64,), device="cuda")
srcs, dsts = L.src_array([L.make_src(x)]), L.dst_array([L.make_dst(out)])
st = L.stream_ptr()
ops = {
    "gsd_conv3x3 (direct taps, general kernel)": lambda: lib.gsd_conv3x3(srcs, 1, wt.data_ptr(), CIN, M, dsts, 1, part.data_ptr(), N, H, W, st),
    **({"gsd_conv3x3_first (store-stream kernel)": lambda: lib.gsd_conv3x3_first(x.data_ptr(), w.data_ptr(), CIN, M, out.data_ptr(), part.data_ptr(), N, H, W, st)}
       if hasattr(lib, "gsd_conv3x3_first") else {}),
}
for name, fn in ops.items():
    for _ in range(2):
        L.check(fn(), name)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        L.check(fn(), name)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"{name:44s} {ms:.4f} ms  ({N * M * H * W * 4 / ms / 1e9:.2f} TB/s of output)", flush=True)
