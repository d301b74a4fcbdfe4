#!/usr/bin/env python3
"""conv3x3 64 -> 64 @320x427, batch N: the DMA-filled kernel (gsd_bf16_conv3x3) against the weights-resident one
(gsd_bf16_conv3x3_c64), forward (statistics epilogue), dX (fused BatchNorm-backward epilogue) and the dX that recomputes
inc's first raw output; each launch timed with HIP events on its own."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gelslim_depth_amd import _lib as L  # noqa: E402

N, H, W, M = int(os.environ.get("N", "32")), 320, 427, 64
lib = L.lib
x = torch.rand((N, 3, H, W), device="cuda")
a = torch.randn((N, H, W, M), device="cuda").to(torch.bfloat16)
yb = torch.randn((N, H, W, M), device="cuda").to(torch.bfloat16)
out = torch.empty_like(a)
mp = lib.gsd_bf16_conv_mpad(M)
w1 = torch.randn((M, M, 3, 3), device="cuda") * 0.05
w0 = torch.randn((M, 3, 3, 3), device="cuda") * 0.3
img = torch.zeros(lib.gsd_bf16_weight_image_size(0, M, M), dtype=torch.bfloat16, device="cuda")
img0 = torch.zeros(lib.gsd_bf16_weight_image_size(2, M, 3), dtype=torch.bfloat16, device="cuda")
L.check(lib.gsd_bf16_weight_image(0, w1.data_ptr(), M, M, img.data_ptr(), L.stream_ptr()), "w")
L.check(lib.gsd_bf16_weight_image(2, w0.data_ptr(), M, 3, img0.data_ptr(), L.stream_ptr()), "w0")
part = torch.empty((4096 * 2 * mp,), device="cuda")
vec = [torch.rand(M, device="cuda") + 0.5, torch.randn(M, device="cuda") * 0.1, torch.randn(M, device="cuda") * 0.1, torch.rand(M, device="cuda") + 0.5]
din, dout, dyb = L.make_nhwc(a), L.make_nhwc(out), L.make_nhwc(yb)
bw = L.gsd_bf16_bnbwd()
bw.y = C.pointer(dyb)
bw.scale, bw.shift, bw.mean, bw.invstd = [v.data_ptr() for v in vec]
st = L.stream_ptr()
ops = {
    "DMA kernel, forward (statistics)": lambda: lib.gsd_bf16_conv3x3(C.byref(din), img.data_ptr(), C.byref(dout), M, M, part.data_ptr(), None, st),
    "c64 kernel, forward (statistics)": lambda: lib.gsd_bf16_conv3x3_c64(C.byref(din), img.data_ptr(), C.byref(dout), part.data_ptr(), None, st),
    "DMA kernel, dX + fused pass 1": lambda: lib.gsd_bf16_conv3x3(C.byref(din), img.data_ptr(), C.byref(dout), M, M, part.data_ptr(), C.byref(bw), st),
    "c64 kernel, dX + fused pass 1": lambda: lib.gsd_bf16_conv3x3_c64(C.byref(din), img.data_ptr(), C.byref(dout), part.data_ptr(), C.byref(bw), st),
    "DMA kernel, plain dX": lambda: lib.gsd_bf16_conv3x3(C.byref(din), img.data_ptr(), C.byref(dout), M, M, None, None, st),
    "first_bn_bwd_reduce (separate pass)": lambda: lib.gsd_bf16_first_bn_bwd_reduce(x.data_ptr(), N, 3, H, W, img0.data_ptr(), C.byref(dout), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), part.data_ptr(), st),
    # (the third epilogue -- pass 1 on y0 recomputed from x, gsd_bf16_conv3x3_c64_dgrad_first -- was measured at 0.80 ms and removed)
}
for name, fn in ops.items():
    for _ in range(2):
        L.check(fn(), name)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.check(fn(), name)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:52s} {e0.elapsed_time(e1) / 5:.4f} ms", flush=True)
