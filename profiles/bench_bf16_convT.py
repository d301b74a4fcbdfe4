#!/usr/bin/env python3
"""The four ConvTranspose2d(k2, s2) of the decoder at batch N in the bf16 engine's form: forward (scatter + bias), dX with the fused
BatchNorm-backward pass 1, dW -- each under GSD_BF16_CTGEMM / GSD_BF16_WGRAD_BIG = 0 (general DMA-filled kernels) and = 1 (large-tile kernels)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gelslim_depth_amd import _lib as L  # noqa: E402

N = int(os.environ.get("N", "32"))
HS, WS = [320, 160, 80, 40, 20], [427, 213, 106, 53, 26]
LEVELS = [(4, 1024), (3, 512), (2, 256), (1, 128)]
lib = L.lib


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tot = {}
for lvl, cin in LEVELS:
    cout, h, w = cin // 2, HS[lvl], WS[lvl]
    H2, W2 = HS[lvl - 1], WS[lvl - 1]
    oy, ox = (H2 - 2 * h) // 2, (W2 - 2 * w) // 2
    x = torch.randn((N, h, w, cin), device="cuda").to(torch.bfloat16)
    wT = torch.randn((cin, cout, 2, 2), device="cuda") / cin ** 0.5
    bias = torch.randn((cout,), device="cuda")
    img_f = torch.empty((lib.gsd_bf16_weight_image_size(3, cout, cin),), dtype=torch.bfloat16, device="cuda")
    img_d = torch.empty((lib.gsd_bf16_weight_image_size(4, cout, cin),), dtype=torch.bfloat16, device="cuda")
    L.check(lib.gsd_bf16_weight_image(3, wT.data_ptr(), cout, cin, img_f.data_ptr(), L.stream_ptr()), "img")
    L.check(lib.gsd_bf16_weight_image(4, wT.data_ptr(), cout, cin, img_d.data_ptr(), L.stream_ptr()), "img")
    cat = torch.zeros((N, H2, W2, 2 * cout), dtype=torch.bfloat16, device="cuda")
    gcat = torch.randn((N, H2, W2, 2 * cout), device="cuda").to(torch.bfloat16)
    y = torch.randn((N, h, w, cin), device="cuda").to(torch.bfloat16)
    dz = torch.empty_like(y)
    coef = [torch.rand((cin,), device="cuda") + 0.5 for _ in range(4)]
    mp = lib.gsd_bf16_conv_mpad(cin)
    part = torch.empty((4096, 2 * mp), device="cuda")
    z = L.int_array([0])
    ty, tx = L.int_array([oy, oy, oy + 1, oy + 1]), L.int_array([ox, ox + 1, ox, ox + 1])
    din, dup, dgy, ddz, dyv = L.make_nhwc(x), L.make_nhwc(cat, cout, cout), L.make_nhwc(gcat, cout, cout), L.make_nhwc(dz), L.make_nhwc(y)
    bw = L.gsd_bf16_bnbwd()
    bw.y = C.pointer(dyv)
    bw.scale, bw.shift, bw.mean, bw.invstd = (t.data_ptr() for t in coef)
    ws_n = lib.gsd_bf16_wgrad_workspace(4, N, h, w, cin, cout)
    ws = torch.empty((ws_n,), device="cuda")
    dw = torch.empty((cin * cout * 4,), device="cuda")
    fl = 2.0 * 4 * cout * cin * N * h * w

    def fwd():
        L.check(lib.gsd_bf16_conv_dense(C.byref(din), img_f.data_ptr(), C.byref(dup), cin, 4 * cout, 1, 1, z, z, h, w, cout, oy, ox,
                                        bias.data_ptr(), None, None, L.stream_ptr()), "fwd")

    def dx():
        L.check(lib.gsd_bf16_conv_dense(C.byref(dgy), img_d.data_ptr(), C.byref(ddz), cout, cin, 4, 2, ty, tx, h, w, 0, 0, 0, None,
                                        part.data_ptr(), C.byref(bw), L.stream_ptr()), "dx")

    def dwf():
        L.check(lib.gsd_bf16_wgrad(C.byref(din), C.byref(dgy), 4, 2, ty, tx, dw.data_ptr(), cout, ws.data_ptr(), ws_n, L.stream_ptr()), "dw")

    line = f"up level {lvl}: {cin:4d} -> {cout:3d} @ {h}x{w}:"
    for what, fn in (("fwd", fwd), ("dX+bn", dx), ("dW", dwf)):
        for v in ("0", "1"):
            os.environ["GSD_BF16_CTGEMM"] = os.environ["GSD_BF16_WGRAD_BIG"] = v
            ms = timed(fn)
            tot[(what, v)] = tot.get((what, v), 0.0) + ms
            line += f"  {what}[{v}] {ms:.3f} ms ({fl / ms / 1e9:.0f} TF)"
    print(line, flush=True)
print("totals: " + "  ".join(f"{k[0]}[{k[1]}] {v:.3f} ms" for k, v in tot.items()))
