"""Run ONE fp32 conv3x3 layer in the two-dimensional Winograd form a few times, the way the engine's forward calls it (deferred
BatchNorm + ReLU source with slack, BatchNorm partial sums), for rocprofv3 --pmc passes on a single shape.
usage: python3 profiles/one_conv_w2d.py ci co h w [B] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
ci, co, h, w = [int(a) for a in sys.argv[1:5]]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
st = L.stream_ptr()
x = L.slack_empty((B, ci, h, w), "cuda")
x.normal_()
sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
y = torch.empty(B, co, h, w, device="cuda")
rows = lib.gsd_conv3x3_w2d_partial_rows(B, h, w, co)
part = torch.zeros(rows * 2 * ((co + 63) // 64 * 64), device="cuda")
src, dst = L.src_array([L.make_src(x, sc, sh, relu=True, slack=L.SLACK)]), L.dst_array([L.make_dst(y)])
wl = torch.empty(lib.gsd_weight_layout_size(8, co, ci), device="cuda")
check(lib.gsd_weight_layout(8, wt.data_ptr(), co, ci, wl.data_ptr(), st), "layout")
for _ in range(reps):
    check(lib.gsd_conv3x3_w2d(src, 1, wl.data_ptr(), ci, co, dst, 1, part.data_ptr(), B, h, w, st), "conv")
torch.cuda.synchronize()
print("done", float(y.abs().mean()), "input MB %.1f output MB %.1f weight image MB %.2f" % (x.numel() * 4e-6, y.numel() * 4e-6, wl.numel() * 4e-6))
