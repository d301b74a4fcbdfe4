"""sha256 of gsd_conv3x3_wgrad's output on three seeded shapes (A/B of bit-identity between two builds of libgsd.so, GSD_LIB_PATH)."""
import hashlib, torch, ctypes as C
from gelslim_depth_amd import _lib as L
lib, check = L.lib, L.check
torch.manual_seed(0)
st = L.stream_ptr()
for (n, k, m, h, w, plain) in [(2, 64, 64, 37, 53, False), (1, 128, 64, 80, 106, True), (2, 32, 96, 21, 19, False), (4, 256, 128, 40, 53, False)]:
    x = L.slack_empty((n, k, h, w), "cuda"); x.normal_()
    sc = torch.rand(k, device="cuda") + 0.5; sh = torch.randn(k, device="cuda") * 0.3
    dy = L.slack_empty((n, m, h, w), "cuda"); dy.normal_()
    dw = torch.empty(m, k, 3, 3, device="cuda")
    need = lib.gsd_conv3x3_wgrad_workspace(n, h, w, k, m)
    ws = torch.empty(max(need, 64), device="cuda")
    s = L.make_src(x, slack=L.SLACK) if plain else L.make_src(x, sc, sh, relu=True, slack=L.SLACK)
    d = L.make_src(dy, slack=L.SLACK)
    check(lib.gsd_conv3x3_wgrad(L.src_array([s]), 1, C.byref(d), k, m, dw.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, st), "wgrad")
    torch.cuda.synchronize()
    print(n, k, m, h, w, plain, hashlib.sha256(dw.cpu().numpy().tobytes()).hexdigest()[:16])
