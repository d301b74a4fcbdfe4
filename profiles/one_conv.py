"""Run ONE fp32 Winograd conv3x3 layer a few times (for rocprofv3 --pmc / --kernel-trace passes on a single kernel).
usage: python3 profiles/one_conv.py ci co h w [B] [reps]"""
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
x = torch.randn(B, ci, h, w, device="cuda")
sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
y = torch.empty(B, co, h, w, device="cuda")
src, dst = L.src_array([L.make_src(x, sc, sh, relu=True)]), L.dst_array([L.make_dst(y)])
wl = torch.empty(lib.gsd_weight_layout_size(4, co, ci), device="cuda")
check(lib.gsd_weight_layout(4, wt.data_ptr(), co, ci, wl.data_ptr(), st), "layout")
for _ in range(reps):
    check(lib.gsd_conv3x3_w43(src, 1, wl.data_ptr(), ci, co, dst, 1, None, B, h, w, st), "conv")
torch.cuda.synchronize()
print("done", float(y.abs().mean()))
