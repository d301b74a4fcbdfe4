"""CPU: the C-ABI library loads here (no GPU) and exports every symbol include/*.h declares; the ctypes
binding lists exactly those symbols.  No compute call is made."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def header_functions():
    names = set()
    inc = os.path.join(REPO, "include")
    for f in sorted(os.listdir(inc)):
        if f.endswith(".h"):
            src = re.sub(r"/\*.*?\*/", "", open(os.path.join(inc, f)).read(), flags=re.S)
            names |= set(re.findall(r"\b(gsd_[a-z0-9_A-Z]+)\s*\(", src))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from gelslim_depth_amd import _lib
    names = header_functions()
    assert len(names) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/*.h but not exported by libgsd.so"
    assert sorted(_lib.SIGNATURES.keys()) == names
    assert "gfx950" in _lib.version()


def test_struct_layout_matches_header():
    from gelslim_depth_amd import _lib
    # gsd_src: 3 pointers, 8 int32 (w_stride + slack), 2 int64 ; gsd_dst: 1 pointer, 6 int32 (w_stride), 2 int64
    assert ctypes.sizeof(_lib.gsd_src) == 3 * 8 + 8 * 4 + 2 * 8
    assert ctypes.sizeof(_lib.gsd_dst) == 8 + 6 * 4 + 2 * 8
    assert _lib.gsd_src.w_stride.offset == 48 and _lib.gsd_src.slack.offset == 52 and _lib.gsd_src.n_stride.offset == 56
    assert _lib.gsd_dst.w_stride.offset == 28 and _lib.gsd_dst.n_stride.offset == 32
    assert ctypes.sizeof(_lib.gsd_nhwc) == 8 + 8 + 4 * 4 and _lib.gsd_nhwc.N.offset == 16
    assert ctypes.sizeof(_lib.gsd_bf16_wimg_job) == 8 + 8 + 4 * 4 and _lib.gsd_bf16_wimg_job.mode.offset == 16


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "gelslim_depth_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_module_refuses_cpu_tensors():
    import pytest
    import torch
    from gelslim_depth_amd.models.unet import UNet
    m = UNet(3, 1, layer_dimensions=[4, 8])
    with pytest.raises(RuntimeError):
        m(x=torch.zeros(1, 3, 8, 8))
    with pytest.raises(NotImplementedError):
        UNet(3, 1, kernel_size=5)
    with pytest.raises(ValueError):
        UNet(3, 1, layer_dimensions=[4, 12])


def test_checkpoint_layout_round_trips_with_the_reference_model(tmp_path):
    """the 118-key state_dict written by the drop-in module loads into the reference's own UNet with
    strict=True and vice versa.  Needs /root/reference (build container only); skipped where it is absent."""
    import sys
    import pytest
    import torch
    if not os.path.isdir("/root/reference/gelslim_depth"):
        pytest.skip("reference tree not present on this machine")
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    try:
        from gelslim_depth.models.unet import UNet as RefUNet
    finally:
        sys.path.remove("/root/reference")
    from gelslim_depth_amd.models.unet import UNet
    dims = [8, 16, 32]
    ours, ref = UNet(3, 1, layer_dimensions=dims), RefUNet(3, 1, layer_dimensions=dims)
    sd_o, sd_r = ours.state_dict(), ref.state_dict()
    assert list(sd_o.keys()) == list(sd_r.keys())
    for k in sd_o:
        assert sd_o[k].shape == sd_r[k].shape and sd_o[k].dtype == sd_r[k].dtype, k
    assert [n for n, _ in ours.named_parameters()] == [n for n, _ in ref.named_parameters()]
    p = tmp_path / "w.pth"
    torch.save(sd_o, p)
    ref.load_state_dict(torch.load(p, map_location="cpu"), strict=True)
    torch.save(ref.state_dict(), p)
    ours.load_state_dict(torch.load(p, map_location="cpu"), strict=True)
    # the reference's init loop (train_unet.py:248-250) touches the same tensors in both
    assert [n for n, _ in ours.named_parameters() if "weight" in n] == [n for n, _ in ref.named_parameters() if "weight" in n]
    # default initialisation statistics match torch's (kaiming-uniform bound = 1/sqrt(fan_in))
    for (n1, p1), (n2, p2) in zip(ours.named_parameters(), ref.named_parameters()):
        if p1.dim() == 4:
            assert abs(p1.abs().max().item() - p2.abs().max().item()) < 0.35 * p2.abs().max().item(), n1


def test_conv_form_choice_and_size_queries(monkeypatch):
    """Host-side planning of the two conv3x3 forms (no kernel is launched): the per-shape choice on the U-Net's layers,
    the work the Winograd form saves, and the workspace / layout queries the engine sizes its buffers with."""
    from gelslim_depth_amd import _lib
    lib = _lib.lib
    monkeypatch.delenv("GSD_CONV_ALGO", raising=False)
    monkeypatch.delenv("GSD_WGRAD_ALGO", raising=False)
    h, w = 320, 427
    for lvl, c in enumerate([64, 128, 256, 512, 1024]):
        cin = 3 if lvl == 0 else c // 2
        for ci, co in ((cin, c), (c, c)):
            algo = lib.gsd_conv3x3_algo(32, h, w, ci, co)
            assert algo == (0 if ci < 16 else 1), (lvl, ci, co, algo)
            direct_mfma = 2.0 * 9 * 32 * h * w * ci * co / 2048          # unpadded direct-tap MFMA count
            w43 = lib.gsd_conv3x3_w43_mfma_count(32, h, w, ci, co)
            if ci >= 16:
                assert 0.5 * direct_mfma <= w43 <= 0.78 * direct_mfma, (lvl, w43 / direct_mfma)   # half + tile padding
            assert lib.gsd_conv3x3_w43_partial_rows(32, h, w, co) > 0
            # Winograd weight images: 18 rows per input channel instead of 9, 64-column m-blocks
            assert lib.gsd_weight_layout_size(4, co, ci) == -(-co // 64) * 64 * (-(-ci // 4) * 4) * 18
            assert lib.gsd_weight_layout_size(5, co, ci) == -(-ci // 64) * 64 * (-(-co // 4) * 4) * 18
            assert lib.gsd_conv3x3_wgrad_workspace(32, h, w, ci, co) > 0
            # two-dimensional Winograd F(2x4,3x3): a third of the direct MFMA work (+ tile padding; no row folding at the deep
            # levels), 24 weight rows per input channel; preferred at batch 32 wherever the model prices it faster, always in
            # eval mode (its bits do not depend on the batch), never for the 3-channel layer
            if ci >= 16:
                w2d = lib.gsd_conv3x3_w2d_mfma_count(32, h, w, ci, co)
                assert direct_mfma / 3 <= w2d <= 0.56 * direct_mfma, (lvl, w2d / direct_mfma)
                assert lib.gsd_conv3x3_w2d_supported(ci, ci) == 1 and lib.gsd_conv3x3_w2d_partial_rows(32, h, w, co) > 0
                assert lib.gsd_weight_layout_size(8, co, ci) == -(-co // 64) * 64 * ci * 24
                assert lib.gsd_weight_layout_size(9, co, ci) == -(-ci // 64) * 64 * co * 24
                assert lib.gsd_conv3x3_prefers_w2d(32, h, w, ci, co, 0) == 1
                t2, t1 = lib.gsd_conv3x3_w2d_estimate_us(32, h, w, ci, co), lib.gsd_conv3x3_w43_estimate_us(32, h, w, ci, co, 1)
                assert t1 > 0 and t2 > 0 and lib.gsd_conv3x3_prefers_w2d(32, h, w, ci, co, 1) == int(t2 < t1)
                if lvl <= 2:
                    assert t2 < t1, (lvl, ci, co, t2, t1)
            else:
                assert lib.gsd_conv3x3_prefers_w2d(32, h, w, ci, co, 0) == 0 and lib.gsd_conv3x3_w2d_supported(ci, ci) == 0
        h, w = h // 2, w // 2
    monkeypatch.setenv("GSD_CONV_ALGO", "0")
    assert lib.gsd_conv3x3_algo(32, 320, 427, 64, 64) == 0
    monkeypatch.setenv("GSD_CONV_ALGO", "1")
    assert lib.gsd_conv3x3_algo(32, 320, 427, 3, 64) == 1


def test_eval_forms_do_not_depend_on_the_batch_and_slab_scratch_is_consistent(monkeypatch):
    """Host-side planning only.  (1) Eval-mode inference promises that image i of a batch gets the bits the image alone gets: the
    conv form the engine takes for an eval forward (gsd_conv3x3_algo + gsd_conv3x3_prefers_w2d(train=0)) must be the same at
    every batch size, for every unit of the U-Net -- and it must be a form whose arithmetic does not depend on the batch (the
    two-dimensional Winograd kernel: no row folding, no K slabs; or direct taps for the 3-channel layer).  (2) The K-slab scratch
    the row form asks for is S x (tile blocks) x 64 x 256 floats with 2 <= S <= 8, and only the small deep levels ask."""
    from gelslim_depth_amd import _lib
    lib = _lib.lib
    for k in ("GSD_CONV_ALGO", "GSD_CONV_W2D", "GSD_W43_SPLIT", "GSD_W2D_SPLIT"):
        monkeypatch.delenv(k, raising=False)
    dims = [64, 128, 256, 512, 1024]
    units = []          # (level, cin, first-segment channels, cout)
    for lvl, c in enumerate(dims):
        cin = 3 if lvl == 0 else dims[lvl - 1]
        units += [(lvl, cin, cin, c), (lvl, c, c, c)]
    for lvl in range(3, -1, -1):
        units += [(lvl, 2 * dims[lvl], dims[lvl], dims[lvl]), (lvl, dims[lvl], dims[lvl], dims[lvl])]
    hs, ws = [320, 160, 80, 40, 20], [427, 213, 106, 53, 26]
    for lvl, cin, c0, cout in units:
        forms = set()
        for n in (1, 2, 3, 8, 16, 32, 64):
            algo = lib.gsd_conv3x3_algo(n, hs[lvl], ws[lvl], cin, cout)
            if algo == 1 and lib.gsd_conv3x3_w2d_supported(cin, c0) and lib.gsd_conv3x3_prefers_w2d(n, hs[lvl], ws[lvl], cin, cout, 0):
                algo = 2
            forms.add(algo)
        assert forms == ({0} if cin < 16 else {2}), (lvl, cin, cout, forms)
    asked = asked2 = 0
    for lvl, cin, c0, cout in units:
        for n in (1, 4, 8, 16, 32):
            for ci, co in ((cin, cout), (cout, cin)):          # forward and dX launches
                if ci < 16:
                    continue
                need = lib.gsd_conv3x3_w43_workspace(n, hs[lvl], ws[lvl], ci, co)
                rows = lib.gsd_conv3x3_w43_partial_rows(n, hs[lvl], ws[lvl], co)
                base = rows // 4 * (-(-co // 64))
                assert need % (base * 64 * 256) == 0 and need // (base * 64 * 256) in (0, 2, 3, 4, 5, 6, 7, 8), (lvl, n, ci, co, need)
                if need:
                    asked += 1
                    assert base < 3 * 512, ("only launches of less than three rounds of the chip's 512 block slots are cut", lvl, n, base)
                # the two-dimensional form: two partial rows per pixel tile; same scratch formula; an eval-mode launch is never cut
                # (gsd_conv3x3_w2d takes no scratch at all), and a cut launch has at least 8 chunks per slab
                if lib.gsd_conv3x3_w2d_supported(ci, ci):
                    need2 = lib.gsd_conv3x3_w2d_workspace(n, hs[lvl], ws[lvl], ci, co)
                    base2 = lib.gsd_conv3x3_w2d_partial_rows(n, hs[lvl], ws[lvl], co) // 2 * (-(-co // 64))
                    s2 = need2 // (base2 * 64 * 256)
                    assert need2 % (base2 * 64 * 256) == 0 and s2 in (0, 2, 3, 4, 5, 6, 7, 8), (lvl, n, ci, co, need2)
                    if need2:
                        asked2 += 1
                        assert base2 < 3 * 512 and ci // 4 // s2 >= 8, (lvl, n, ci, co, base2, s2)
    assert asked > 0 and asked2 > 0


def test_guard_struct_and_bench_self_launch_refuses_cleanly():
    """gsd_guard layout; and `python bench.py --gpus N` without a rank environment starts its own ranks as a child job --
    on a machine with fewer than N GPUs (this container has none) it must say so and exit non-zero without a traceback."""
    import subprocess
    import sys
    from gelslim_depth_amd import _lib
    assert ctypes.sizeof(_lib.gsd_guard) == 16 and _lib.gsd_guard.tick.offset == 8
    import torch
    if torch.cuda.device_count() >= 2:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "one per GPU" in r.stderr and "Traceback" not in r.stderr and r.stdout.strip() == ""


def test_wgrad_w2d_kernels_use_no_scratch():
    """gsd_wgrad_w2d.hip lives at the 256-register limit of two waves per SIMD.  Its first form spilled, and hipcc stored two spill
    slots on some paths of a branchy prologue only while reloading them on every path (wild global addresses, a memory fault at
    the 160x213 level).  The kernels are written to need no scratch; this keeps it that way (device-only compile, ~6 s)."""
    import re
    import shutil
    import subprocess
    from gelslim_depth_amd import build as b
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc on this machine")
    r = subprocess.run([hipcc] + b.CFLAGS + [f"-I{b.INCLUDE}", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c",
                        os.path.join(b.CSRC, "gsd_wgrad_w2d.hip"), "-o", os.devnull], capture_output=True, text=True, check=True)
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    kernels = {n: s for n, s in zip(names, scratch) if "wgrad3x3_w2d_kernel" in n}
    assert len(kernels) >= 2, r.stderr[-2000:]
    assert all(v == 0 for v in kernels.values()), kernels
