#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config: train frames/s at 3x320x427, batch 32 per GPU,
fp32, full train step (forward + MSE + backward + Adam + EMA), every device operation a libgsd HIP kernel.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one synthetic batch already resident in HBM.  Data parallel,
weak scaling: every rank processes its own batch of 32; gradients are all-reduced over RCCL.
Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for how each field is obtained).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

DIMS = [64, 128, 256, 512, 1024]
H, W = 320, 427
FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0       # same guide: bf16 MFMA dense peak (not the 2:1-sparsity figure)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE.json: 32)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: fixed GLOBAL batch split over the ranks (configs[3]: 64); overrides --batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--per-layer", action="store_true", help="print per-shape conv3x3 TFLOP/s to stderr")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: the metric's arithmetic (BASELINE configs[1-3], default); bf16: mixed precision, configs[4]")
    ap.add_argument("--graph", action="store_true", help="--workload infer only: replay the forward as one hipGraph")
    ap.add_argument("--workload", choices=["train", "infer"], default="train",
                    help="train: BASELINE configs[2] (the metric, default); infer: configs[1], eval-mode forward only")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1 or os.environ.get("GSD_FORCE_SYNC"):
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
        dist.init_process_group(backend="nccl", device_id=dev)     # "nccl" is RCCL on ROCm
        pg = dist.group.WORLD

    from gelslim_depth_amd import synth
    from gelslim_depth_amd.models.unet import UNet
    from gelslim_depth_amd.train import TrainStep

    model = UNet(n_channels=3, n_classes=1, layer_dimensions=DIMS, precision="bf16" if args.dtype == "bf16" else "fp32")
    st = synth.make_state(3, 1, DIMS, 0, "conditioned")            # random-init weights of the named architecture
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
    model = model.to(dev).train()
    step = TrainStep(model, lr=1e-3, weight_decay=1e-6, ema_decay=0.995, loss="mse", process_group=pg,
                     sync_bn=args.sync_bn)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    B = args.batch
    if args.global_batch > 0:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be divisible by the number of ranks")
        B = args.global_batch // world
    x = torch.rand((B, 3, H, W), device=dev, generator=g)                  # U[0,1)  (post /255 difference image)
    tgt = -0.9 * torch.rand((B, 1, H, W), device=dev, generator=g)         # U(-0.9,0] (normalised depth)

    def barrier():
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.workload == "infer":
        model.eval()
        if args.graph:
            from gelslim_depth_amd.graph import GraphedInference
            graphed = GraphedInference(model, x)

            def one_step():
                graphed(x)
        else:
            def one_step():
                with torch.no_grad():
                    model(x=x)
    else:
        def one_step():
            step(x, tgt)
    for _ in range(args.warmup):
        one_step()
    barrier()
    eng = model._engine
    eng.kernel_log = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    klog = eng.kernel_log
    eng.kernel_log = None
    loss = float(step.last_loss.item())
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        # dominant kernel: the conv3x3 kernel (forward + dX launches) with the largest share of the timed region.
        # klog rows: (kernel, algorithmic flops, event, event, shape[, flops executed by the MFMAs])
        klog = [r if len(r) > 5 else tuple(r) + (r[1],) for r in klog]
        peak = FP32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
        by_kernel = {}
        for v, f, a, b, _, ex in klog:
            d = by_kernel.setdefault(v, [0.0, 0.0, 0, 0.0])
            d[0] += f
            d[1] += a.elapsed_time(b)
            d[2] += 1
            d[3] += ex
        if args.dtype == "f32":
            dom = max(by_kernel, key=lambda k: by_kernel[k][1]) if by_kernel else "conv3x3_w43_kernel"
        else:
            dom = "bf16_conv3x3"
        flops, ms, launches, executed = by_kernel.get(dom, [0.0, 0.0, 0, 0.0])
        all_flops = sum(r[1] for r in klog)
        all_ms = sum(r[2].elapsed_time(r[3]) for r in klog)
        if args.per_layer:
            import collections
            agg = collections.OrderedDict()
            for v, f, a, b, sig, _ in klog:
                d = agg.setdefault(sig, [0.0, 0.0, 0])
                d[0] += f
                d[1] += a.elapsed_time(b)
                d[2] += 1
            if args.dtype == "bf16":
                byname = collections.OrderedDict()
                for v, f, a, b, _, _ in klog:
                    d = byname.setdefault(v, [0.0, 0.0, 0])
                    d[0] += f
                    d[1] += a.elapsed_time(b)
                    d[2] += 1
                for v, (f, t, c) in byname.items():
                    print("%-16s launches %4d total %.2f ms/step  %.1f TFLOP/s" %
                          (v, c, t / args.steps, f / (t * 1e-3) / 1e12), file=sys.stderr)
            else:
                for sig, (f, t, c) in agg.items():
                    print("conv3x3 M%-5d K%-5d %3dx%-3d launches %3d avg %.3f ms  %.1f TFLOP/s" %
                          (sig + (c, t / c, f / (t * 1e-3) / 1e12)), file=sys.stderr)
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic = None      # HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json)
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
            if tj.get("kernel") == dom and B == 32 and args.dtype == "f32":
                traffic = round(float(tj["hbm_bytes_per_launch"]))
        except (OSError, ValueError, KeyError):
            pass
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "train frames/sec (320x427) at batch 32" if args.workload == "train"
                      else "inference frames/sec (320x427), eval-mode forward",
            "value": round(B * world * args.steps / elapsed, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.global_batch > 0 else "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": (("BASELINE.json configs[2]: batch-%d full train step (fwd+MSE+bwd+Adam+EMA) fp32, " % B
                                     if args.dtype == "f32" else
                                     "BASELINE.json configs[4] (per-GPU share): batch-%d full train step, bf16 mixed precision "
                                     "(bf16 NHWC activations on bf16 MFMA, fp32 master weights/statistics/Adam), " % B) +
                                    "3x320x427 -> 1x320x427, U-Net [64,128,256,512,1024], all HIP kernels")
                                   if args.workload == "train" else
                                   ("BASELINE.json configs[1]: eval-mode forward fp32, 3x320x427 -> 1x320x427, "
                                    "U-Net [64,128,256,512,1024], HIP kernels"),
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "sync_bn": bool(args.sync_bn), "final_loss": round(loss, 6),
                       "conv3x3_form": ("bf16 MFMA implicit GEMM" if args.dtype == "bf16" else
                                        {"0": "direct taps", "1": "winograd F(4,3) rows"}.get(
                                            os.environ.get("GSD_CONV_ALGO", ""), "winograd F(4,3) rows, fp32 (Cin>=16) / direct taps (first layer)"))},
            "roofline": {"bound": "mfma", "kernel": dom,
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         # achieved = ALGORITHMIC flops (2*9*Cout*Cin per pixel) / time.  The Winograd F(4,3) kernel executes
                         # half of them (+ tile padding), so frac can exceed 1; mfma_frac = flops its MFMAs actually issued
                         # / time / peak is the matrix-core utilisation.
                         "mfma_executed": round(executed / (ms * 1e-3) / 1e12, 2) if ms > 0 else 0.0,
                         "mfma_frac": round(executed / (ms * 1e-3) / 1e12 / peak, 4) if ms > 0 else 0.0,
                         "launches_timed": launches, "avg_launch_ms": round(ms / max(launches, 1), 4),
                         "gflop_per_launch": round(flops / max(launches, 1) / 1e9, 2),
                         "all_conv3x3_tflops": round(all_flops / (all_ms * 1e-3) / 1e12, 2) if all_ms > 0 else 0.0,
                         "conv3x3_share_of_step": round(all_ms / args.steps / ms_per_step, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
