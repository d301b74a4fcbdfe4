"""Build profiles/traffic.json from three rocprofv3 --pmc passes over `bench.py --steps 1 --warmup 0` (FETCH_SIZE; WRITE_SIZE;
SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE), as written by profiles/run_pmc.sh.  All template variants of the dominant
kernel are pooled.  FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request: MI355X_MICROARCH.md, HBM).
usage: python profiles/make_traffic.py <tag> [kernel]     (reads gpurun_out/<tag>_{fetch,write,sq}_pmc.txt, profiles named per round;
kernel: the dominant kernel's name, default conv3x3_w2d_kernel).  The JSON records the source digest of the library the passes ran
(gelslim_depth_amd/csrc/libgsd.so.stamp): bench.py marks the figure stale when it differs from the library it runs."""
import json
import re
import sys

tag = sys.argv[1]
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "conv3x3_w2d_kernel"
OUT = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"


def lib_digest():
    try:
        return open("gelslim_depth_amd/csrc/libgsd.so.stamp").read().strip()
    except OSError:
        return None


def pooled(path, counters):
    tot = {c: 0.0 for c in counters}
    n = 0
    cur = None
    for line in open(path):
        m = re.match(r"^(\S.*) dispatches (\d+)$", line.rstrip())
        if m:
            cur = (m.group(1), int(m.group(2))) if KERNEL in m.group(1) else None
            if cur:
                n += cur[1]
            continue
        if cur:
            m = re.match(r"^\s+(\S+)\s+total (\S+)", line)
            if m and m.group(1) in tot:
                tot[m.group(1)] += float(m.group(2))
    return {c: v / max(n, 1) for c, v in tot.items()}, n


def algorithmic_bytes_per_launch(n=32):
    """SURVEY 8(d)-style count for the dominant kernel's 34 launches of a step (17 forward + 17 dX, Cin >= 16): every source and
    destination tensor once (4 bytes per element) + the 24-row weight image; a dX launch of a DoubleConv's second conv also reads
    the producer's raw output (fused BatchNorm backward).  -> (average, forward average, dX average) in bytes."""
    hs, ws, cs = [320, 160, 80, 40, 20], [427, 213, 106, 53, 26], [64, 128, 256, 512, 1024]
    layers = [(0, 64, 64, True)]
    for l in range(1, 5):
        layers += [(l, cs[l - 1], cs[l], False), (l, cs[l], cs[l], True)]
    for l in (3, 2, 1, 0):
        layers += [(l, 2 * cs[l], cs[l], False), (l, cs[l], cs[l], True)]
    fw = dx = 0
    for l, ci, co, second in layers:
        px = n * hs[l] * ws[l]
        fw += 4 * px * (ci + co) + 96 * ci * co
        dx += 4 * px * (ci + co) + 96 * ci * co + (4 * px * ci if second else 0)
    return (fw + dx) / 34.0, fw / 17.0, dx / 17.0


def wgrad_algorithmic_bytes_per_launch(n=32):
    """conv3x3 dW, 17 launches: the activation and the gradient once each + the (Cout, Cin, 3, 3) result."""
    hs, ws, cs = [320, 160, 80, 40, 20], [427, 213, 106, 53, 26], [64, 128, 256, 512, 1024]
    layers = [(0, 64, 64)]
    for l in range(1, 5):
        layers += [(l, cs[l - 1], cs[l]), (l, cs[l], cs[l])]
    for l in (3, 2, 1, 0):
        layers += [(l, 2 * cs[l], cs[l]), (l, cs[l], cs[l])]
    return sum(4 * n * hs[l] * ws[l] * (ci + co) + 36 * ci * co for l, ci, co in layers) / 17.0


fetch, n1 = pooled(f"gpurun_out/{tag}_fetch_pmc.txt", ["FETCH_SIZE"])
write, n2 = pooled(f"gpurun_out/{tag}_write_pmc.txt", ["WRITE_SIZE"])
sq, n3 = pooled(f"gpurun_out/{tag}_sq_pmc.txt", ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"])
cycles = sq["GRBM_GUI_ACTIVE"] / 8.0                      # rocprofv3 sums the 8 XCDs
busy = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0) if cycles else None   # 1024 SIMDs
out = {
    "kernel": KERNEL,
    "workload": f"bench.py --steps 1 --warmup 0 (batch 32), {n1} dispatches averaged, all template variants pooled",
    "FETCH_SIZE_KB_per_launch": fetch["FETCH_SIZE"],
    "WRITE_SIZE_KB_per_launch": write["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM)",
    "hbm_bytes_per_launch": (2 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0,
    "algorithmic_bytes_per_launch": algorithmic_bytes_per_launch()[0] if "conv3x3" in KERNEL else wgrad_algorithmic_bytes_per_launch(),
    "algorithmic_bytes_forward_dX": list(algorithmic_bytes_per_launch()[1:]) if "conv3x3" in KERNEL else None,
    "mfma_busy": busy,
    "cycles_per_launch": cycles,
    "sources": [f"profiles/{tag}_pmc_fetch_size.txt", f"profiles/{tag}_pmc_write_size.txt", f"profiles/{tag}_pmc_sq.txt"],
    "library_source_digest": lib_digest(),
}
json.dump(out, open(OUT, "w"), indent=1)
print(json.dumps(out, indent=1))
