"""First-layer dW (gsd_conv3x3_wgrad_bn: BatchNorm backward applied on the fly) at the U-Net's size, against the pair it
replaces (gsd_bn_bwd_apply in place + gsd_conv3x3_wgrad).  usage (GPU box): PYTHONPATH=. python profiles/bench_wgrad_first.py [batch]"""
import ctypes as C
import sys
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ci, co, h, w = 3, 64, 320, 427
st = L.stream_ptr()
x = torch.randn(B, ci, h, w, device="cuda")
dz, raw = torch.randn(B, co, h, w, device="cuda"), torch.randn(B, co, h, w, device="cuda")
par = [torch.rand(co, device="cuda") + 0.5 for _ in range(5)]
dw, dw2 = torch.empty(co, ci, 3, 3, device="cuda"), torch.empty(co, ci, 3, 3, device="cuda")
need = max(lib.gsd_conv3x3_wgrad_bn_workspace(B, h, w, ci, co), lib.gsd_conv3x3_wgrad_workspace(B, h, w, ci, co))
ws = torch.empty(need, device="cuda")
a = L.make_src(x)


def fused():
    check(lib.gsd_conv3x3_wgrad_bn(C.byref(a), dz.data_ptr(), raw.data_ptr(), *[p.data_ptr() for p in par], ci, co, dw.data_ptr(),
                                   ws.data_ptr(), need, B, h, w, st), "wgrad_bn")


g = dz.clone()


def pair():
    check(lib.gsd_bn_bwd_apply(g.data_ptr(), raw.data_ptr(), *[p.data_ptr() for p in par], B, co, h, w, None, 0, st), "apply")
    dy = L.make_src(g)
    check(lib.gsd_conv3x3_wgrad(L.src_array([a]), 1, C.byref(dy), ci, co, dw2.data_ptr(), ws.data_ptr(), need, B, h, w, st), "wgrad")


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    return best


tf = timeit(fused)
gb = 2 * dz.numel() * 4 / 1e9
print("fused dW (reads dz + raw = %.2f GB): %.3f ms = %.2f TB/s" % (gb, tf, gb / tf))
g.copy_(dz)
pair()
fused()
torch.cuda.synchronize()
print("pair vs fused rel diff %.2e" % ((dw - dw2).abs().sum() / dw2.abs().sum()).item())
print("apply + general dW: %.3f ms" % timeit(pair))
