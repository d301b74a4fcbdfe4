"""K-slab sweep of the Winograd conv3x3 (gsd_conv3x3_w43_ws) over the (M, K, H, W) launches of one fp32 train step: forward and
dX shapes of the U-Net's levels, at a per-GPU batch.  One line per launch shape: ms unsplit, ms with S = 2..8 slabs forced, the
planner's own choice and its time.  Calibrates w43_pick_slabs (gsd_conv3x3_w43.hip).
usage (GPU box): python profiles/bench_kslabs.py [batch] [min level]"""
import os
import sys
import torch
from gelslim_depth_amd import _lib as L

lib, check = L.lib, L.check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
LMIN = int(sys.argv[2]) if len(sys.argv) > 2 else 1
HS, WS = [320, 160, 80, 40, 20], [427, 213, 106, 53, 26]
C = [64, 128, 256, 512, 1024]
shapes = []          # (level, M, K) as launched: forward c0 / c1, dX c1 / c0, decoder c0 forward + its dX
for lvl in range(1, 5):
    shapes += [(lvl, C[lvl], C[lvl - 1]), (lvl, C[lvl], C[lvl]), (lvl, C[lvl - 1], C[lvl])]
for lvl in range(0, 4):
    shapes += [(lvl, C[lvl], 2 * C[lvl]), (lvl, 2 * C[lvl], C[lvl])]
shapes = sorted(set(s for s in shapes if s[0] >= LMIN))
st = L.stream_ptr()


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot_plain = tot_auto = 0.0
for lvl, m, k in shapes:
    h, w = HS[lvl], WS[lvl]
    x = L.slack_empty((B, k, h, w), "cuda")
    x.normal_()
    wt = torch.randn(m, k, 3, 3, device="cuda") * 0.05
    y = torch.empty(B, m, h, w, device="cuda")
    rows = lib.gsd_conv3x3_w43_partial_rows(B, h, w, m)
    part = torch.zeros(rows * 2 * ((m + 63) // 64 * 64), device="cuda")
    wl = torch.empty(lib.gsd_weight_layout_size(4, m, k), device="cuda")
    check(lib.gsd_weight_layout(4, wt.data_ptr(), m, k, wl.data_ptr(), st), "layout")
    src, dst = L.src_array([L.make_src(x, slack=L.SLACK)]), L.dst_array([L.make_dst(y)])
    os.environ["GSD_W43_SPLIT"] = "8"
    cap = max(lib.gsd_conv3x3_w43_workspace(B, h, w, k, m), 1)
    for s_ in range(2, 8):
        os.environ["GSD_W43_SPLIT"] = str(s_)
        cap = max(cap, lib.gsd_conv3x3_w43_workspace(B, h, w, k, m))
    ws = torch.empty(cap, device="cuda")
    res = {}
    for s_ in ["0", "2", "3", "4", "5", "6", "8", "auto"]:
        if s_ == "auto":
            os.environ.pop("GSD_W43_SPLIT", None)
        else:
            os.environ["GSD_W43_SPLIT"] = s_
        need = lib.gsd_conv3x3_w43_workspace(B, h, w, k, m)
        if s_ not in ("0", "auto") and need == 0:
            continue
        res[s_] = (timed(lambda: check(lib.gsd_conv3x3_w43_ws(src, 1, wl.data_ptr(), k, m, dst, 1, part.data_ptr(), ws.data_ptr(),
                                                             cap, B, h, w, st), "conv")), need)
    os.environ.pop("GSD_W43_SPLIT", None)
    gf = 2.0 * 9 * B * h * w * k * m / 1e9
    auto_ms, auto_need = res["auto"]
    base = rows // 4 * ((m + 63) // 64)
    auto_s = auto_need // (base * 64 * 256) if auto_need else 1
    tot_plain += res["0"][0]
    tot_auto += auto_ms
    print("L%d M%-4d K%-4d %3dx%-3d blocks %5d | " % (lvl, m, k, h, w, base) +
          " ".join("S%s %.3f" % (s_ if s_ != "0" else "1", v[0]) for s_, v in res.items() if s_ != "auto") +
          " | auto S=%d %.3f ms %.0f TF (unsplit %.0f TF)" % (auto_s, auto_ms, gf / auto_ms, gf / res["0"][0]), flush=True)
print("sum over shapes: unsplit %.3f ms, planner %.3f ms" % (tot_plain, tot_auto))
