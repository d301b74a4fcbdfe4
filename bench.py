#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: train frames/s at 3x320x427, batch 32 per GPU,
fp32, full train step (forward + MSE + backward + Adam + EMA), every device operation a libgsd HIP kernel.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: this process starts N ranks of itself through
`python -m torch.distributed.run` (a CHILD process, before anything here has touched the GPU), forwards rank 0's
JSON line and exits with the child's code.  Under torch.distributed.run (RANK / WORLD_SIZE set) it is one rank.

A "step" is one pass of the hot path over one synthetic batch already resident in HBM.  Data parallel,
weak scaling: every rank processes its own batch of 32; gradients are all-reduced over RCCL.
Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for how each field is obtained).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

DIMS = [64, 128, 256, 512, 1024]
H, W = 320, 427
FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0       # same guide: bf16 MFMA dense peak (not the 2:1-sparsity figure)
BF16_MFMA_SUSTAINED_TFLOPS = 1855.5  # measured: the ideal bf16 MFMA loop at the clock the chip holds under it (r04_mfma_shape_ubench.txt)
HBM_PEAK_TBS = 8.0                   # same guide: HBM3E spec peak (6.3 TB/s is what a copy kernel reaches)
# SURVEY.md 8(d), the north-star kernel `inc` (3->64->64 @320x427), ideal-fusion fp32 bytes per image: read x 1.64 MB,
# write + re-read raw c0 34.98 + 34.98, write raw c1 34.98  (the consumer's re-read of c1 belongs to the next kernel)
INC_ALGO_BYTES_PER_IMAGE_FP32 = (3 + 3 * 64) * H * W * 4


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE.json: 32)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: fixed GLOBAL batch split over the ranks (configs[3]: 64); overrides --batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the short extra legs at N=1 (configs[1] inference, configs[3] shares at batch 16 / 8, bf16 at 16 / 32)")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the 'val L1 vs ref' leg at N=1 (one eval forward against tests/golden/gfull_b1.npz, outside the timed region)")
    ap.add_argument("--dry-run-ranks", type=int, default=0,
                    help="first-contact rehearsal of the N-rank flow on a box with fewer GPUs: N ranks share GPU 0 over gloo at tiny "
                         "dimensions, the parent checks rank 0's line (n_ranks_seen, per-rank ms, allreduce fields, strong leg) and that "
                         "no other rank writes to stdout.  Timings are meaningless; at most 6 ranks (the GPU box's process guard)")
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--per-layer", action="store_true", help="print per-shape conv3x3 TFLOP/s to stderr")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: the metric's arithmetic (BASELINE configs[1-3], default); bf16: mixed precision, configs[4]")
    ap.add_argument("--graph", action="store_true", help="--workload infer only: replay the forward as one hipGraph")
    ap.add_argument("--workload", choices=["train", "infer"], default="train",
                    help="train: BASELINE configs[2] (the metric, default); infer: configs[1], eval-mode forward only")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` with no rank environment: run the N ranks as a child torch.distributed.run job.
    Nothing in THIS process has initialised the GPU (torch is not even imported yet), so no exec of a GPU process occurs."""
    import torch      # device_count() does not initialise the GPU on this image
    have = torch.cuda.device_count()
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} asks for {args.gpus} ranks, one per GPU, but this node shows {have} GPU(s); "
              "RCCL refuses two ranks on one device. Run on a node with enough GPUs.", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)        # launcher / RCCL chatter never reaches stdout
    if line is not None:
        print(line, flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        return 1
    return proc.returncode


def dry_run_ranks(n: int) -> int:
    """`--dry-run-ranks N`: start the N-rank job exactly as the driver does (python -m torch.distributed.run ... bench.py --gpus N),
    but with the ranks sharing GPU 0 over gloo (GSD_BENCH_BACKEND / GSD_BENCH_SHARE_GPU) and tiny dimensions (GSD_BENCH_TINY), and
    check what the first real multi-GPU run will be judged on.  Prints one JSON line with the checks; exit code 0 iff all hold."""
    if n < 2 or n > 6:
        print("bench.py: --dry-run-ranks wants 2..6 ranks (the GPU box allows 6 processes on its card)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(n), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    env = dict(os.environ, GSD_BENCH_BACKEND="gloo", GSD_BENCH_SHARE_GPU="1", GSD_BENCH_TINY="1", OMP_NUM_THREADS="2")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out_lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    checks = {"exit_code_0": proc.returncode == 0,
              "stdout_is_one_json_line": len(out_lines) == 1 and out_lines[0].startswith("{")}
    rec = {}
    if checks["stdout_is_one_json_line"]:
        try:
            rec = json.loads(out_lines[0])
        except ValueError:
            checks["stdout_is_one_json_line"] = False
    ar = rec.get("allreduce") or {}
    strong = (rec.get("extra") or {}).get("configs[3] strong scaling: global batch 64 fp32") or {}
    checks.update({
        "n_gpus_and_n_ranks_seen": rec.get("n_gpus") == n and rec.get("n_ranks_seen") == n,
        "per_rank_ms_per_step": isinstance(rec.get("per_rank_ms_per_step"), list) and len(rec.get("per_rank_ms_per_step")) == n,
        "value_is_whole_job": _whole_job(rec),
        "scaling_weak": rec.get("scaling") == "weak",
        "allreduce_fields": all(k in ar for k in ("allreduce_ms_per_step", "exposed_ms_per_step", "bus_GBps", "MB_per_step",
                                                  "per_rank_allreduce_ms")) and len(ar.get("per_rank_allreduce_ms", [])) == n,
        "strong_leg": (64 % n != 0) or (strong.get("scaling") == "strong" and strong.get("n_gpus") == n
                                        and strong.get("global_batch") == 64 and "allreduce" in strong),
        "labelled_rehearsal": "REHEARSAL" in (rec.get("config") or {}).get("workload", ""),
    })
    ok = all(checks.values())
    print(json.dumps({"dry_run_ranks": n, "ok": ok, "checks": checks, "rank0_line": rec,
                      "stdout_lines_seen": [ln[:200] for ln in out_lines] if not ok else len(out_lines),
                      "stderr_tail": proc.stderr.splitlines()[-5:] if not ok else []}), flush=True)
    return 0 if ok else 1


def host_cores() -> int:
    """CPU cores this process may actually use: min(affinity mask, cgroup quota, logical CPUs)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(seconds_budget: float = 25.0):
    """The reference's CPU path (its torch-operator sequence, oracle/torch_cpu_path.py) timed on this box's host
    cores on a bounded sample of the same workload: full train steps at 320x427, batch 2."""
    import numpy as np
    import torch
    from gelslim_depth_amd import synth
    from oracle import torch_cpu_path as ot      # cpu_baseline leg only
    cores = host_cores()
    torch.set_num_threads(cores)
    st = synth.make_state(3, 1, DIMS, 0, "conditioned")
    b = 2
    x, t = synth.make_batch(b, H, W, 1)
    tr = ot.CpuTrainer(st)
    xt, tt = torch.from_numpy(x), torch.from_numpy(t)
    tr.step(xt, tt)                               # warm-up
    times = []
    t_start = time.time()
    while len(times) < 3 or (time.time() - t_start < seconds_budget and len(times) < 5):
        t0 = time.time()
        tr.step(xt, tt)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return {"value": round(b / med, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} full train steps (fwd+MSE+bwd+Adam) of batch {b} at 3x{H}x{W} fp32, "
                      f"torch CPU operators as the reference executes them, median {med:.2f} s/step"}


def traffic_provenance(tj):
    """How a committed rocprofv3 --pmc figure relates to the library this run loads: the JSON carries the source digest of the
    library the passes ran (profiles/make_traffic.py); `stale` says that the sources have changed since (the figure may still
    hold -- most changes do not touch the kernel -- but nobody has re-measured it)."""
    try:
        from gelslim_depth_amd import build as b
        cur = b.source_digest()
    except Exception:      # noqa: BLE001 -- provenance only
        cur = None
    rec = tj.get("library_source_digest")
    return {"profiled_library_digest": rec[:16] if rec else None, "this_library_digest": cur[:16] if cur else None,
            "stale": bool(rec is None or cur is None or rec != cur)}


def parity_vs_reference(dev, dtype):
    """BASELINE.json's "val L1 vs ref" (the path of /root/reference/test_utils/test_depth_estimation.py:17,61-65: build the
    model, load weights, eval(), model(x=...)): one eval-mode forward of the full-size net at 1x3x320x427 through the HIP path,
    against the output the REFERENCE itself produced for the same weights and input on PyTorch-CPU (tests/golden/gfull_b1.npz,
    made by tests/golden/make_golden.py importing the reference; weights and input are regenerated from the fixture's seed by
    the build-owned generators).  Runs outside the timed region.  Bound: the north star's 1e-3 relative L1 on fp32 depth;
    the bf16 mixed-precision form is not the metric's arithmetic and is reported against the 2e-2 its test allows."""
    import numpy as np
    import torch
    from gelslim_depth_amd import synth
    from gelslim_depth_amd.models.unet import UNet
    fixture = "gfull_b1.npz"
    g = np.load(os.path.join(REPO, "tests", "golden", fixture))
    dims = [int(v) for v in g["meta/dims"]]
    seed = int(g["meta/seed"])
    st = synth.make_state(3, 1, dims, seed, "conditioned")
    x, _ = synth.make_batch(1, H, W, seed + 1)
    m = UNet(n_channels=3, n_classes=1, layer_dimensions=dims, precision="bf16" if dtype == "bf16" else "fp32")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
    m = m.to(dev).eval()
    with torch.no_grad():
        y = m(x=torch.from_numpy(x).to(dev)).double().cpu().numpy()
    ref = g["y_eval"].astype(np.float64)
    rel = float(np.abs(y - ref).sum() / np.abs(ref).sum())
    bound = 1e-3 if dtype == "f32" else 2e-2
    out = {"rel_l1": float("%.3e" % rel), "bound": bound, "ok": bool(rel <= bound), "fixture": fixture,
           "what": "eval-mode forward, 1x3x320x427, full-size net: sum|y_hip - y_ref| / sum|y_ref| against the reference's own "
                   "PyTorch-CPU output (fixture generated by importing the reference, tests/golden/make_golden.py)",
           "dtype": dtype}
    if dtype == "f32":
        # The timed region is a TRAIN step: one batch-1 train-mode forward + MSE + backward of the same model against what the
        # reference computed (train_utils/train_unet.py:346-347,370,374): loss, the output's checksums, the L2 norm of EVERY
        # parameter gradient, the BatchNorm running statistics the forward updated.  Whole-network gradient bound 2e-2 (a
        # ReLU / max-pool net is chaotic in the last bits; unit-by-unit the kernels hold 1e-5 / 5e-5:
        # tests/test_gpu_net.py::test_backward_teacher_forced_unit_by_unit).
        _, tgt = synth.make_batch(1, H, W, seed + 1)
        m.train()
        yt = m(x=torch.from_numpy(x).to(dev))
        loss = torch.mean((yt - torch.from_numpy(tgt).to(dev)) ** 2)
        loss.backward()
        y64 = yt.detach().double()
        ysum = g["y_train_sum"]
        y_rel = max(abs(y64.sum().item() - ysum[0]) / abs(ysum[0]), abs(y64.abs().sum().item() - ysum[1]) / abs(ysum[1]))
        loss_rel = abs(loss.item() - float(g["loss"])) / abs(float(g["loss"]))
        worst, worst_k = 0.0, None
        for k, p in m.named_parameters():
            want = float(g[f"gradsum/{k}"][2])
            dev_k = abs(p.grad.double().pow(2).sum().sqrt().item() - want) / max(want, 1e-30)
            if dev_k > worst:
                worst, worst_k = dev_k, k
        sd = m.state_dict()
        buf_worst = 0.0
        for k in g.files:
            if k.startswith("bufsum/"):
                d = sd[k[len("bufsum/"):]].double()
                buf_worst = max(buf_worst, abs(d.sum().item() - g[k][0]) / max(abs(g[k][1]), 1e-30))
        tb = {"loss": 2e-4, "y_sums": 2e-4, "grad_l2": 2e-2, "bn_buffers": 1e-3}
        train = {"loss_rel": float("%.3e" % loss_rel), "y_train_sums_rel": float("%.3e" % y_rel),
                 "grad_l2_worst": float("%.3e" % worst), "grad_l2_worst_param": worst_k,
                 "bn_running_stats_worst": float("%.3e" % buf_worst), "bound": tb,
                 "ok": bool(loss_rel <= tb["loss"] and y_rel <= tb["y_sums"] and worst <= tb["grad_l2"] and buf_worst <= tb["bn_buffers"]),
                 "what": "one batch-1 train-mode forward + MSE + backward vs the reference's loss, output checksums, per-parameter "
                         "gradient L2 norms (64 tensors) and BatchNorm running statistics (same fixture)"}
        out["eval"] = {"rel_l1": out["rel_l1"], "bound": bound, "ok": out["ok"]}
        out["train"] = train
        out["ok"] = bool(out["ok"] and train["ok"])
    del m
    torch.cuda.empty_cache()
    return out


class Leg:
    """One timed workload: model + step + synthetic batch resident in HBM."""

    def __init__(self, dev, rank, dtype, workload, batch, pg=None, sync_bn=False, graph=False):
        import torch
        from gelslim_depth_amd import synth
        from gelslim_depth_amd.models.unet import UNet
        from gelslim_depth_amd.train import TrainStep
        self.torch = torch
        self.dtype, self.workload, self.B = dtype, workload, batch
        model = UNet(n_channels=3, n_classes=1, layer_dimensions=DIMS, precision="bf16" if dtype == "bf16" else "fp32")
        st = synth.make_state(3, 1, DIMS, 0, "conditioned")            # random-init weights of the named architecture
        model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
        self.model = model.to(dev).train()
        # the production path: GradSync(timing=False).  The collectives are timed in a short pass of their own (Leg.comm)
        self.step = TrainStep(self.model, lr=1e-3, weight_decay=1e-6, ema_decay=0.995, loss="mse", process_group=pg,
                              sync_bn=sync_bn)
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank)
        self.x = torch.rand((batch, 3, H, W), device=dev, generator=g)                  # U[0,1)  (post /255 difference image)
        self.tgt = -0.9 * torch.rand((batch, 1, H, W), device=dev, generator=g)         # U(-0.9,0] (normalised depth)
        if workload == "infer":
            self.model.eval()
            if graph:
                from gelslim_depth_amd.graph import GraphedInference
                graphed = GraphedInference(self.model, self.x)
                self.one_step = lambda: graphed(self.x)
            else:
                self.one_step = self._infer
        else:
            self.one_step = lambda: self.step(self.x, self.tgt)

    def _infer(self):
        with self.torch.no_grad():
            self.model(x=self.x)

    def run(self, steps, warmup, barrier):
        """W untimed steps, then exactly K steps between two barrier+synchronize; returns (seconds, kernel log, regions)."""
        for _ in range(warmup):
            self.one_step()
        barrier()
        eng = self.model._engine
        # The bf16 engine runs its weight gradients on a side stream, beside the backward chain: events around a kernel of the
        # main stream then time it WITH a co-runner.  Its timed region is therefore run without the per-kernel events, and the
        # kernel durations come from a short pass of their own behind it, in which every kernel has the chip to itself
        # (engine_bf16._on_side).  The fp32 engine is single-stream: its events sit in the timed region itself.
        overlapped = bool(getattr(eng, "side_dw", False)) and self.workload == "train"
        if not overlapped:
            eng.kernel_log, eng.region_log = [], []
            if hasattr(eng, "wgrad_log"):
                eng.wgrad_log = []
        t0 = time.perf_counter()
        for _ in range(steps):
            self.one_step()
        barrier()
        elapsed = time.perf_counter() - t0
        self.logged_steps = steps
        if overlapped:
            self.logged_steps = min(steps, 5)
            eng.kernel_log, eng.region_log = [], []
            if hasattr(eng, "wgrad_log"):
                eng.wgrad_log = []
            for _ in range(self.logged_steps):
                self.one_step()
            barrier()
        klog, rlog = eng.kernel_log, eng.region_log
        self.wlog = getattr(eng, "wgrad_log", None) or []
        eng.kernel_log = eng.region_log = None
        if hasattr(eng, "wgrad_log"):
            eng.wgrad_log = None
        return elapsed, klog, rlog

    def comm(self, barrier, steps=3):
        """Gradient all-reduce time (HIP events on a side stream that waits for RCCL's stream around every bucket,
        gelslim_depth_amd/distributed.py): per step, and the part the compute stream had to wait for.  Measured in `steps`
        extra steps BEHIND the timed region with GradSync's instrumented path switched on -- the metric itself runs the
        production path (no side-stream hop, no events)."""
        sync = getattr(self.step, "sync", None)
        if sync is None:
            return None
        sync.set_timing(True)
        for _ in range(steps):
            self.one_step()
        barrier()
        t = sync.pop_timing()
        sync.set_timing(False)
        n = max(1, t["steps"])
        ms = t["allreduce_ms"] / n
        return {"allreduce_ms_per_step": round(ms, 4), "exposed_ms_per_step": round(t["exposed_ms"] / n, 4),
                "MB_per_step": round(t["bytes"] / n / 1e6, 3),
                "bus_GBps": round(t["bytes"] / n / 1e9 / (ms * 1e-3), 2) if ms > 0 else None,
                "buckets": t["buckets"], "steps_timed": n}

    def inc_hbm(self, rlog):
        """North-star second roofline: the `inc` double-conv forward against the HBM roof.  achieved = ALGORITHMIC bytes
        (SURVEY 8(d) ideal fusion, halved for bf16 storage) / measured time of the region (HIP events on the launch stream)."""
        ms = [a.elapsed_time(b) for name, a, b in rlog if name == "inc_forward"]
        if not ms:
            return None
        avg = sum(ms) / len(ms)
        by = INC_ALGO_BYTES_PER_IMAGE_FP32 * self.B * (0.5 if self.dtype == "bf16" else 1.0)
        tbs = by / (avg * 1e-3) / 1e12
        # the bytes the block REALLY moves: not observed by this run -- the committed rocprofv3 --pmc passes over the block on
        # its own (profiles/inc_block.py, batch 32; FETCH_SIZE x 2 + WRITE_SIZE, summed over its kernels)
        traffic = traffic_src = None
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "inc_traffic.json")))["bf16" if self.dtype == "bf16" else "fp32"]
            traffic = round(float(tj["hbm_bytes_per_run"]) * self.B / 32.0)
            traffic_src = "profiles/inc_traffic.json (committed rocprofv3 --pmc passes: %s), scaled by batch / 32" % ", ".join(tj["sources"])
        except (OSError, ValueError, KeyError):
            pass
        extra = {}
        if traffic:
            extra = {"traffic": traffic, "traffic_source": traffic_src, "traffic_provenance": traffic_provenance(tj),
                     "traffic_vs_algorithmic": round(traffic / by, 3), "traffic_TBps": round(traffic / (avg * 1e-3) / 1e12, 3)}
        return {**extra, "bound": "hbm", "kernel": "inc double-conv forward (3->64->64 @320x427): "
                + ("statistics-only first conv + fused kernel (rebuilds relu(bn(conv(x))) per halo tile, weights-resident 64->64 "
                   "conv) + BN finalize x2 + BN-apply with the max-pool (gsd_bf16_inc.hip)"
                   if (self.dtype == "bf16" and getattr(self.model._engine, "fused_inc", False)) else
                   "first conv straight from x + 64->64 conv + BN statistics + 2 BN-apply launches" if self.dtype == "bf16"
                   else "direct conv + Winograd conv + BN statistics launches"),
                "achieved": round(tbs, 3), "peak": HBM_PEAK_TBS, "unit": "TB/s", "frac": round(tbs / HBM_PEAK_TBS, 4),
                "algorithmic_bytes": int(by), "avg_ms": round(avg, 4), "launches_timed": len(ms),
                "note": ("fp32: this block is MFMA-bound (AI 99 FLOP/B vs ridge 20), the HBM fraction is reported, not targeted"
                         if self.dtype == "f32" else "bf16: AI ~200 FLOP/B vs ridge ~310, HBM roof applies")}


def conv_roofline(klog, dtype, steps, ms_per_step, per_layer=False):
    """Roofline of the dominant conv3x3 kernel from the live HIP-event log.
    klog rows: (kernel, algorithmic flops, event, event, shape[, flops executed by the MFMAs])."""
    klog = [r if len(r) > 5 else tuple(r) + (r[1],) for r in klog]
    peak = FP32_MFMA_PEAK_TFLOPS if dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
    by_kernel = {}
    for v, f, a, b, _, ex in klog:
        d = by_kernel.setdefault(v, [0.0, 0.0, 0, 0.0])
        d[0] += f
        d[1] += a.elapsed_time(b)
        d[2] += 1
        d[3] += ex
    if dtype == "f32":
        dom = max(by_kernel, key=lambda k: by_kernel[k][1]) if by_kernel else "conv3x3_w43_kernel"
    else:
        dom = "bf16_conv3x3"
    flops, ms, launches, executed = by_kernel.get(dom, [0.0, 0.0, 0, 0.0])
    all_flops = sum(r[1] for r in klog)
    all_ms = sum(r[2].elapsed_time(r[3]) for r in klog)
    if per_layer:
        import collections
        if dtype == "bf16":
            for v, (f, t, c, _) in by_kernel.items():
                print("%-16s launches %4d total %.2f ms/step  %.1f TFLOP/s" %
                      (v, c, t / steps, f / (t * 1e-3) / 1e12), file=sys.stderr)
            agg = collections.OrderedDict()
            for v, f, a, b, sig, _ in klog:
                if sig is None:
                    continue
                d = agg.setdefault((v, sig), [0.0, 0.0, 0])
                d[0] += f
                d[1] += a.elapsed_time(b)
                d[2] += 1
            for (v, sig), (f, t, c) in agg.items():
                print("%s M%-5d K%-5d %3dx%-3d launches %3d avg %.3f ms  %.1f TFLOP/s" %
                      ((v,) + tuple(sig) + (c, t / c, f / (t * 1e-3) / 1e12)), file=sys.stderr)
        else:
            agg = collections.OrderedDict()
            for v, f, a, b, sig, _ in klog:
                d = agg.setdefault(sig, [0.0, 0.0, 0])
                d[0] += f
                d[1] += a.elapsed_time(b)
                d[2] += 1
            for sig, (f, t, c) in agg.items():
                print("conv3x3 M%-5d K%-5d %3dx%-3d launches %3d avg %.3f ms  %.1f TFLOP/s" %
                      (sig + (c, t / c, f / (t * 1e-3) / 1e12)), file=sys.stderr)
    algorithmic = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    done = executed / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    # achieved = flops the kernel's MFMA instructions EXECUTE (2048 per v_mfma_f32_16x16x4_f32, tile padding included) / time:
    # <= peak by construction.  The Winograd F(4,3) kernel needs half the multiplications of the direct convolution, so the
    # ALGORITHMIC rate (2*9*Cout*Cin flops per pixel / time, what SURVEY 8(d) prices) is reported next to it and may exceed peak.
    roof = {}
    if dtype != "f32":
        # the NOMINAL peak assumes 2.4 GHz; under a saturated bf16 MFMA stream this chip holds 1.74-1.92 GHz and the ideal loop
        # (all operand reads ahead, no fills) reaches 1.73-1.86 PFLOP/s: profiles/ubench/mfma_bf16_rate32.hip, r04_mfma_shape_ubench.txt
        roof = {"sustained_peak": BF16_MFMA_SUSTAINED_TFLOPS, "frac_of_sustained": round(done / BF16_MFMA_SUSTAINED_TFLOPS, 4),
                "sustained_source": "profiles/r04_mfma_shape_ubench.txt (ideal MFMA loop at the clock the chip holds under it)"}
    return dom, {"bound": "mfma", "kernel": dom, "achieved": round(done, 2), "peak": peak, "unit": "TFLOP/s",
                 "frac": round(done / peak, 4), **roof,
                 "algorithmic_tflops": round(algorithmic, 2), "algorithmic_vs_peak": round(algorithmic / peak, 4),
                 "launches_timed": launches, "avg_launch_ms": round(ms / max(launches, 1), 4),
                 "gflop_per_launch": round(flops / max(launches, 1) / 1e9, 2),
                 "all_conv3x3_tflops": round(all_flops / (all_ms * 1e-3) / 1e12, 2) if all_ms > 0 else 0.0,
                 "conv3x3_share_of_step": round(all_ms / steps / ms_per_step, 3) if ms_per_step > 0 else 0.0}


def _whole_job(rec) -> bool:
    """value == global batch / step time (a malformed record is a failed check, not a traceback)."""
    try:
        v, gb, ms = float(rec.get("value")), float((rec.get("config") or {}).get("global_batch")), float(rec.get("ms_per_step"))
        return ms > 0 and abs(v - gb * 1e3 / ms) <= 1e-2 * v
    except (TypeError, ValueError):
        return False


def second_roofline(wlog, steps, ms_per_step):
    """The second kernel of the step -- conv3x3 dW (launch + its ordered slab reduction) -- priced like the dominant one: `achieved` =
    flops its MFMAs EXECUTE / time (<= peak by construction), `algorithmic_*` = 2*9*Cout*Cin flops per pixel / time."""
    if not wlog:
        return None
    by = {}
    for v, f, a, b, _, ex in wlog:
        d = by.setdefault(v, [0.0, 0.0, 0, 0.0])
        d[0] += f
        d[1] += a.elapsed_time(b)
        d[2] += 1
        d[3] += ex
    dom = max(by, key=lambda k: by[k][1])
    flops, ms, launches, executed = by[dom]
    all_ms = sum(d[1] for d in by.values())
    return {"bound": "mfma", "kernel": dom + " (+ wgrad_w43_reduce_kernel)", "achieved": round(executed / (ms * 1e-3) / 1e12, 2),
            "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(executed / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "algorithmic_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
            "algorithmic_vs_peak": round(flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "launches_timed": launches, "avg_launch_ms": round(ms / launches, 4), "gflop_per_launch": round(flops / launches / 1e9, 2),
            "ms_per_step": round(all_ms / steps, 3), "share_of_step": round(all_ms / steps / ms_per_step, 3) if ms_per_step > 0 else 0.0}


def workload_name(dtype, workload, B):
    if workload == "infer":
        return ("BASELINE.json configs[1]: batch-%d eval-mode forward %s, 3x320x427 -> 1x320x427, "
                "U-Net [64,128,256,512,1024], HIP kernels" % (B, "fp32" if dtype == "f32" else "bf16"))
    if dtype == "f32":
        head = "BASELINE.json configs[2]: batch-%d full train step (fwd+MSE+bwd+Adam+EMA) fp32, " % B
    else:
        head = (("BASELINE.json configs[4] (per-GPU share of 128 on 8 GPUs): " if B == 16 else
                 "bf16 mixed-precision form of configs[2] (configs[4]'s arithmetic at the metric's batch): ") +
                "batch-%d full train step, bf16 mixed precision "
                "(bf16 NHWC activations on bf16 MFMA, fp32 master weights/statistics/Adam), " % B)
    return head + "3x320x427 -> 1x320x427, U-Net [64,128,256,512,1024], all HIP kernels"


def main():
    args = parse_args()
    if args.dry_run_ranks:
        sys.exit(dry_run_ranks(args.dry_run_ranks))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if os.environ.get("GSD_BENCH_TINY"):       # rehearsal dimensions (--dry-run-ranks): a small net on small images
        global DIMS, H, W
        DIMS, H, W = [16, 32, 64], 40, 53

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    # Rehearsal of the multi-rank flow on a box with fewer GPUs than ranks (timings are then meaningless): GSD_BENCH_BACKEND=gloo
    # GSD_BENCH_SHARE_GPU=1 lets the ranks share device 0 over gloo; the measured configuration is always one rank per GPU on RCCL
    backend = os.environ.get("GSD_BENCH_BACKEND", "nccl")
    share = bool(os.environ.get("GSD_BENCH_SHARE_GPU")) and backend != "nccl"
    if local_rank >= torch.cuda.device_count() and not share:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU of its own ({torch.cuda.device_count()} visible); "
                         "one rank per GPU is required (RCCL refuses duplicate devices)")
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pg = None
    n_ranks_seen = 1
    def init_group():
        import torch.distributed as dist
        # keep stdout to the ONE JSON line: RCCL writes its version banner (NCCL_DEBUG=VERSION is exported in this image) and
        # its warnings to stdout; send whatever it has to say to stderr instead
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            del os.environ["NCCL_DEBUG"]
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # ... and the C++ side of a backend may write to fd 1 while it connects (gloo: "[Gloo] Rank r is connected to ..."): the
        # process's stdout points at stderr for the duration of the rendezvous
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=dev)     # "nccl" is RCCL on ROCm
            else:
                dist.init_process_group(backend=backend)
                dist.barrier()
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        return dist.group.WORLD

    if world > 1 or os.environ.get("GSD_FORCE_SYNC"):
        pg = init_group()
        n_ranks_seen = torch.distributed.get_world_size()

    B = args.batch
    if args.global_batch > 0:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be divisible by the number of ranks")
        B = args.global_batch // world

    def barrier():
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def over_ranks(local_elapsed, comm):
        """(max elapsed over ranks, [per-rank ms per step], rank-max of the collective times) -- rank 0 reports them."""
        if pg is None:
            return local_elapsed, [round(local_elapsed / args.steps * 1e3, 3)], comm
        tab = torch.zeros((n_ranks_seen, 3), device=dev, dtype=torch.float64)     # a gather as a sum: every rank fills its own row
        tab[rank] = torch.tensor([local_elapsed, comm["allreduce_ms_per_step"] if comm else 0.0,
                                  comm["exposed_ms_per_step"] if comm else 0.0], dtype=torch.float64)
        torch.distributed.all_reduce(tab)
        tab = tab.cpu()
        if comm:
            comm = dict(comm, allreduce_ms_per_step=round(float(tab[:, 1].max()), 4),
                        exposed_ms_per_step=round(float(tab[:, 2].max()), 4),
                        per_rank_allreduce_ms=[round(float(v), 4) for v in tab[:, 1]])
        return float(tab[:, 0].max()), [round(float(v), 3) for v in tab[:, 0] / args.steps * 1e3], comm

    leg = Leg(dev, rank, args.dtype, args.workload, B, pg=pg, sync_bn=args.sync_bn, graph=args.graph)
    elapsed, klog, rlog = leg.run(args.steps, args.warmup, barrier)
    loss = float(leg.step.last_loss.item())
    elapsed, per_rank_ms, comm = over_ranks(elapsed, leg.comm(barrier))
    hbm = leg.inc_hbm(rlog) if rank == 0 else None
    logged_steps = leg.logged_steps
    main_wlog = getattr(leg, "wlog", None)

    # configs[3] names a batch-64 scaling sweep without saying whether 64 is global or per GPU (report both):
    # the metric above is the weak-scaling line (fixed per-GPU batch); this short leg is the STRONG-scaling one -- the global
    # batch of 64 split over the ranks -- run by every rank of the same invocation, after the metric's timed region
    strong = None
    if args.workload == "train" and args.dtype == "f32" and not args.no_extra and args.global_batch == 0 and 64 % world == 0:
        del leg
        torch.cuda.empty_cache()
        sb, ssteps, swarm = 64 // world, 5, 2
        lg = Leg(dev, rank, "f32", "train", sb, pg=pg, sync_bn=args.sync_bn)
        el, _, _ = lg.run(ssteps, swarm, barrier)
        sc = lg.comm(barrier)
        if pg is not None:
            t2 = torch.tensor([el, sc["allreduce_ms_per_step"] if sc else 0.0, sc["exposed_ms_per_step"] if sc else 0.0],
                              device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t2, op=torch.distributed.ReduceOp.MAX)
            el = float(t2[0])
            if sc:
                sc = dict(sc, allreduce_ms_per_step=round(float(t2[1]), 4), exposed_ms_per_step=round(float(t2[2]), 4))
        strong = {"workload": "BASELINE.json configs[3]: batch-64 data-parallel train step fp32 (GLOBAL batch 64 = %d per GPU x %d), "
                              "3x320x427, all HIP kernels, gradients summed over RCCL" % (sb, world),
                  "value": round(64 * ssteps / el, 2), "unit": "frames/s", "scaling": "strong", "n_gpus": world,
                  "per_gpu_batch": sb, "global_batch": 64, "ms_per_step": round(el / ssteps * 1e3, 3), "steps": ssteps,
                  "warmup": swarm, "dtype": "f32", "final_loss": round(float(lg.step.last_loss.item()), 6)}
        if sc is not None:
            strong["allreduce"] = sc
        del lg
        torch.cuda.empty_cache()
        leg = None

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        dom, roof = conv_roofline(klog, args.dtype, logged_steps, ms_per_step, args.per_layer)
        # HBM bytes per launch of the dominant kernel: NOT observed by this run -- read from the committed rocprofv3 --pmc passes
        # (separate FETCH_SIZE / WRITE_SIZE runs, gfx950 correction applied; profiles/traffic.json names its sources)
        roof["traffic"], roof["traffic_source"] = None, None
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
            if tj.get("kernel") == dom and B == 32 and args.dtype == "f32":
                roof["traffic"] = round(float(tj["hbm_bytes_per_launch"]))
                roof["traffic_source"] = "profiles/traffic.json (committed rocprofv3 --pmc passes: %s)" % ", ".join(tj.get("sources", []))
                roof["traffic_provenance"] = traffic_provenance(tj)
                # algorithmic bytes of an average launch of the dominant kernel: the source segments and the
                # destination once each + the weight image (profiles/make_traffic.py): forward 0.98 GB, dX 1.23 GB, average 1.105 GB
                roof["algorithmic_bytes_per_launch"] = int(tj.get("algorithmic_bytes_per_launch", 1.105e9))
                roof["traffic_vs_algorithmic"] = round(roof["traffic"] / roof["algorithmic_bytes_per_launch"], 3)
        except (OSError, ValueError, KeyError):
            pass
        if args.dtype == "f32" and args.workload == "train":
            sec = second_roofline(main_wlog, logged_steps, ms_per_step)
            if sec is not None:
                try:
                    tw = json.load(open(os.path.join(REPO, "profiles", "traffic_wgrad.json")))
                    if tw.get("kernel", "?") in sec["kernel"] and B == 32:
                        sec["traffic"] = round(float(tw["hbm_bytes_per_launch"]))
                        sec["algorithmic_bytes_per_launch"] = int(tw["algorithmic_bytes_per_launch"])
                        sec["traffic_vs_algorithmic"] = round(sec["traffic"] / sec["algorithmic_bytes_per_launch"], 3)
                        sec["mfma_busy_pmc"] = round(float(tw["mfma_busy"]), 3) if tw.get("mfma_busy") else None
                        sec["traffic_source"] = "profiles/traffic_wgrad.json (committed rocprofv3 --pmc passes: %s)" % ", ".join(tw.get("sources", []))
                        sec["traffic_provenance"] = traffic_provenance(tw)
                except (OSError, ValueError, KeyError):
                    pass
                roof["second"] = sec
        if hbm is not None:
            roof["hbm"] = hbm
        out = {
            "metric": "train frames/sec (320x427) at batch 32" if args.workload == "train"
                      else "inference frames/sec (320x427), eval-mode forward",
            "value": round(B * world * args.steps / elapsed, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "n_ranks_seen": n_ranks_seen,
            "per_rank_ms_per_step": per_rank_ms,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.global_batch > 0 else "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": workload_name(args.dtype, args.workload, B) +
                                   ("" if backend == "nccl" else f" [REHEARSAL over {backend}, ranks sharing a GPU: {share} -- not a measurement]"),
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "sync_bn": bool(args.sync_bn), "final_loss": round(loss, 6),
                       "conv3x3_form": ("bf16 MFMA implicit GEMM" if args.dtype == "bf16" else
                                        {"0": "direct taps", "1": "winograd, fp32"}.get(
                                            os.environ.get("GSD_CONV_ALGO", ""),
                                            "fp32: two-dimensional winograd F(2x4,3x3) or winograd F(4,3) rows with K slabs, per launch by "
                                            "the library's run-time models (Cin>=16) / direct taps (first layer)"))},
            "roofline": roof,
            # which machine measured this line: the pool's boxes differ by 2-3 %; the dominant kernel's average
            # launch time in `roofline` (1.43-1.44 ms on the fast group, ~1.48 on the slow one) tells the group
            "box": {"hostname": __import__("socket").gethostname(), "device": torch.cuda.get_device_name(dev_index),
                    "compute_units": torch.cuda.get_device_properties(dev_index).multi_processor_count},
        }
        if comm is not None:
            comm["share_of_step"] = round(comm["allreduce_ms_per_step"] / ms_per_step, 4) if ms_per_step > 0 else None
            comm["exposed_share_of_step"] = round(comm["exposed_ms_per_step"] / ms_per_step, 4) if ms_per_step > 0 else None
            out["allreduce"] = comm
        extra = {}
        if strong is not None:
            extra["configs[3] strong scaling: global batch 64 fp32"] = strong
        if world == 1 and not args.no_extra and args.workload == "train" and args.dtype == "f32":
            # the other single-GPU configurations BASELINE.json names, a few steps each, AFTER the timed region of the metric
            leg = None
            torch.cuda.empty_cache()
            # configs[3]'s strong-scaling endpoints on ONE GPU: the per-GPU share of the global batch of 64 on 4 and on 8 GPUs
            # (16, 8) -- t(B64) / (t(B8) + ring all-reduce) bounds what 8 GPUs can give before any multi-GPU box is available;
            # configs[4]'s per-GPU share is 128 / 8 = 16 images in bf16; the bf16 step at batch 32 is the bf16 twin of the metric
            legs = (("configs[1] batch-16 inference fp32", "f32", "infer", 16),
                    ("configs[3] per-GPU share on 4 GPUs: batch-16 fp32 train step", "f32", "train", 16),
                    ("configs[3] per-GPU share on 8 GPUs: batch-8 fp32 train step", "f32", "train", 8),
                    ("configs[4] per-GPU share: batch-16 bf16 train step", "bf16", "train", 16),
                    ("bf16 twin of the metric: batch-32 bf16 train step", "bf16", "train", 32))
            for key, dt, wl, b in legs:
                lg = Leg(dev, 0, dt, wl, b)
                el, kl, rl = lg.run(5, 2, barrier)
                _, rf = conv_roofline(kl, dt, lg.logged_steps, el / 5 * 1e3)
                e = {"workload": workload_name(dt, wl, b), "value": round(b * 5 / el, 2), "unit": "frames/s",
                     "ms_per_step": round(el / 5 * 1e3, 3), "steps": 5, "warmup": 2, "dtype": dt, "roofline": rf}
                if getattr(lg.model._engine, "side_dw", False) and wl == "train":
                    e["roofline"]["timing"] = ("kernel durations from %d extra steps behind the timed ones, every kernel alone on the chip; "
                                               "the timed steps run the weight gradients on a side stream" % lg.logged_steps)
                h = lg.inc_hbm(rl)
                if h is not None:
                    e["roofline"]["hbm"] = h
                if wl == "train":
                    e["final_loss"] = round(float(lg.step.last_loss.item()), 6)
                extra[key] = e
                del lg
                torch.cuda.empty_cache()
            if strong is not None:
                # what one GPU says about the 8-GPU strong-scaling target (north star: >= 6x on batch 64): the 8-GPU step is at
                # best t(B8) + the ring all-reduce of 124 MB over xGMI (~1.4 ms when nothing of it is overlapped)
                t64 = strong["ms_per_step"]
                t16 = extra["configs[3] per-GPU share on 4 GPUs: batch-16 fp32 train step"]["ms_per_step"]
                t8 = extra["configs[3] per-GPU share on 8 GPUs: batch-8 fp32 train step"]["ms_per_step"]
                # ... and what a data-parallel RANK pays on top of the plain step: the same batch-8 step with the gradient buckets
                # live -- RCCL on ONE rank (GSD_FORCE_SYNC): nine bucket hand-offs on the hand-off stream, nine all-reduce calls, the
                # final wait -- everything of the 8-GPU step but the wire.  The bound is computed from THIS leg.
                t8s, t8s_err = None, None
                try:
                    os.environ["GSD_FORCE_SYNC"] = "1"
                    pg1 = init_group()
                    lg = Leg(dev, 0, "f32", "train", 8, pg=pg1)
                    el, _, _ = lg.run(5, 2, barrier)
                    t8s = round(el / 5 * 1e3, 3)
                    del lg
                    torch.cuda.empty_cache()
                    torch.distributed.destroy_process_group()
                except Exception as e:      # noqa: BLE001 -- the bound then falls back to the plain step, and says so
                    t8s_err = repr(e)
                finally:
                    os.environ.pop("GSD_FORCE_SYNC", None)
                t8b = t8s if t8s is not None else t8
                extra["configs[3] strong-scaling bound from one GPU"] = {
                    "t_B64_ms": t64, "t_B16_ms": t16, "t_B8_ms": t8, "t_B8_one_rank_rccl_ms": t8s,
                    "bucket_handoff_cost_ms": round(t8s - t8, 3) if t8s is not None else None, "one_rank_rccl_error": t8s_err,
                    "assumed_ring_allreduce_ms": 1.4,
                    "speedup_bound_4_gpus": round(t64 / (t16 + 1.4), 3), "speedup_bound_8_gpus": round(t64 / (t8b + 1.4), 3),
                    "bound_uses": "t_B8_one_rank_rccl_ms" if t8s is not None else "t_B8_ms (the one-rank RCCL leg failed)",
                    "per_image_ms": {"B64": round(t64 / 64, 4), "B32": round(ms_per_step / 32, 4) if B == 32 else None,
                                     "B16": round(t16 / 16, 4), "B8": round(t8 / 8, 4)},
                    "note": "single-GPU measurements; an upper bound on the multi-GPU speedup, not a scaling measurement"}
        if extra:
            out["extra"] = extra
        parity_failed = False
        if world == 1 and not args.no_parity:
            leg = None
            torch.cuda.empty_cache()
            # (guarded: whatever goes wrong in a checker leg, the measured line is still printed -- then the run exits non-zero)
            try:
                out["parity"] = parity_vs_reference(dev, args.dtype)
            except Exception as e:      # noqa: BLE001
                out["parity"] = {"ok": False, "error": repr(e)}
            parity_failed = not out["parity"]["ok"]
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:      # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
        if parity_failed:
            print("bench.py: the HIP path's output differs from the reference's by more than the bound: %r" % (out["parity"],),
                  file=sys.stderr)
            sys.exit(3)
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
