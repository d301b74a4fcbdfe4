"""sha256 of gsd_conv3x3_w43's output and statistics on seeded shapes, with and without K slabs (A/B of bit-identity between two
builds of libgsd.so, GSD_LIB_PATH)."""
import hashlib, torch, ctypes as C
from gelslim_depth_amd import _lib as L
lib, check = L.lib, L.check
torch.manual_seed(0)
st = L.stream_ptr()
for (n, k, m, h, w, plain) in [(2, 64, 64, 37, 53, False), (1, 128, 64, 80, 106, True), (2, 30, 96, 21, 19, False), (8, 512, 512, 20, 26, False)]:
    x = L.slack_empty((n, k, h, w), "cuda"); x.normal_()
    wt = torch.randn(m, k, 3, 3, device="cuda") * 0.1
    sc = torch.rand(k, device="cuda") + 0.5; sh = torch.randn(k, device="cuda") * 0.3
    img = torch.empty(lib.gsd_weight_layout_size(4, m, k), device="cuda")
    check(lib.gsd_weight_layout(4, wt.data_ptr(), m, k, img.data_ptr(), st), "wl")
    y = torch.empty(n, m, h, w, device="cuda")
    part = torch.zeros(lib.gsd_conv3x3_w43_partial_rows(n, h, w, m) * 2 * ((m + 63) // 64 * 64), device="cuda")
    need = lib.gsd_conv3x3_w43_workspace(n, h, w, k, m)
    ws = torch.empty(max(need, 64), device="cuda")
    s = L.make_src(x, slack=L.SLACK) if plain else L.make_src(x, sc, sh, relu=True, slack=L.SLACK)
    check(lib.gsd_conv3x3_w43_ws(L.src_array([s]), 1, img.data_ptr(), k, m, L.dst_array([L.make_dst(y)]), 1, part.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, st), "w43")
    torch.cuda.synchronize()
    print(n, k, m, h, w, plain, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha256(part.cpu().numpy().tobytes()).hexdigest()[:16])
