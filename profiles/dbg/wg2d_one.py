"""one dW launch of the 2-D Winograd form on a small shape (debug aid): python profiles/dbg/wg2d_one.py co ci h w n"""
import ctypes as C, sys, torch
from gelslim_depth_amd import _lib as L
co, ci, h, w, n = [int(v) for v in sys.argv[1:6]]
x = L.slack_empty((n, ci, h, w), "cuda"); x.normal_()
p = (w + 3) // 4 * 4
dyb = torch.zeros((n, co, h, p), device="cuda"); dyb[..., :w].normal_()
dy = dyb[..., :w]
dw = torch.empty(co, ci, 3, 3, device="cuda")
need = L.lib.gsd_conv3x3_wgrad_workspace(n, h, w, ci, co)
ws = torch.empty(need, device="cuda")
a, d = L.src_array([L.make_src(x, slack=L.SLACK)]), L.make_src(dy)
print("form", L.lib.gsd_conv3x3_wgrad_form(a, 1, C.byref(d), ci, co, n, h, w), flush=True)
rc = L.lib.gsd_conv3x3_wgrad(a, 1, C.byref(d), ci, co, dw.data_ptr(), ws.data_ptr(), need, n, h, w, L.stream_ptr())
print("rc", rc, L.lib.gsd_last_error(), flush=True)
torch.cuda.synchronize()
ref = torch.nn.grad.conv2d_weight(x, (co, ci, 3, 3), dy.contiguous(), padding=1)
print("rel", ((dw - ref).abs().sum() / ref.abs().sum()).item())
