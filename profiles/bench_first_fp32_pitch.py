#!/usr/bin/env python3
"""A/B: the fp32 first convolution (3 -> 64 @320x427, gsd_conv3x3 direct taps = conv3x3_dma_kernel<1,4,4>) writing
its 64 output planes with a row pitch of 427 (dense, rows never line-aligned), 428 (16-byte aligned rows), 432 (64 B), 448 (128 B)
floats, and the level-0 Winograd 64 -> 64 convolution reading that buffer (dword halo gathers at pitch 427, aligned 16-byte pieces at
the others).  usage (GPU box): python profiles/bench_first_fp32_pitch.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gelslim_depth_amd import _lib as L  # noqa: E402

N, H, W, M, CIN = int(os.environ.get("N", "32")), 320, 427, 64, 3
lib = L.lib
st = L.stream_ptr()
x = torch.rand((N, CIN, H, W), device="cuda")
w = torch.randn((M, CIN, 3, 3), device="cuda") * 0.3
w2 = torch.randn((M, M, 3, 3), device="cuda") * 0.05
wt = torch.zeros(lib.gsd_weight_layout_size(0, M, CIN), device="cuda")
L.check(lib.gsd_weight_layout(0, w.data_ptr(), M, CIN, wt.data_ptr(), st), "layout")
wt2 = torch.zeros(lib.gsd_weight_layout_size(4, M, M), device="cuda")
L.check(lib.gsd_weight_layout(4, w2.data_ptr(), M, M, wt2.data_ptr(), st), "layout")
rows = max(lib.gsd_conv3x3_partial_rows(N, H, W, M), lib.gsd_conv3x3_w43_partial_rows(N, H, W, M), 2048)
part = torch.empty((rows * 2 * 64,), device="cuda")
sc, sh = torch.rand(M, device="cuda") + 0.5, torch.randn(M, device="cuda") * 0.1
ref = None
for pitch in (427, 428, 432, 448, 512, 427):       # dense first AND last: the first configuration of a process runs cold
    base = torch.full((N * M * H * pitch + 64,), float("nan"), device="cuda")   # pad columns: the NaN sentinel a ReLU'd source wants there
    out = base[16:16 + N * M * H * pitch].view(N, M, H, pitch)[..., :W]      # 64-byte aligned start, 4+ floats of slack either side
    y2 = torch.empty((N, M, H, W), device="cuda")
    srcs, dsts = L.src_array([L.make_src(x)]), L.dst_array([L.make_dst(out)])
    src2, dst2 = L.src_array([L.make_src(out, sc, sh, relu=True, slack=4)]), L.dst_array([L.make_dst(y2)])
    ops = {
        "first conv (direct taps) -> pitch %d" % pitch: lambda: lib.gsd_conv3x3(srcs, 1, wt.data_ptr(), CIN, M, dsts, 1, part.data_ptr(), N, H, W, st),
        "Winograd 64->64 reading pitch %d" % pitch: lambda: lib.gsd_conv3x3_w43(src2, 1, wt2.data_ptr(), M, M, dst2, 1, part.data_ptr(), N, H, W, st),
    }
    for name, fn in ops.items():
        for _ in range(2):
            L.check(fn(), name)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(6):
            L.check(fn(), name)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 6
        print(f"{name:44s} {ms:.4f} ms", flush=True)
    if ref is None:
        ref = (out.clone(), y2.clone())
    else:
        print("    bit-identical to the dense run: first conv %s, Winograd output %s" % (torch.equal(out, ref[0]), torch.equal(y2, ref[1])), flush=True)
