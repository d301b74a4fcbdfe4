"""profiles/inc_traffic.json from the two rocprofv3 --pmc passes over profiles/inc_block.py (FETCH_SIZE; WRITE_SIZE) that
profiles/r03_profile.sh writes: real HBM bytes of the `inc` double-conv forward per run, all of its kernels summed (the
output conv of the one-level network left out).  FETCH_SIZE is doubled (gfx950: 64 B counted per 128-B request,
MI355X_MICROARCH.md).  usage: python profiles/make_inc_traffic.py <tag> <fp32|bf16> <runs per process>"""
import json
import re
import sys

tag, prec, runs = sys.argv[1], sys.argv[2], int(sys.argv[3])
SKIP = ("conv1x1_out", "at::native", "__amd_rocclr", "weight_image", "weight_layout")


def total(path, counter):
    tot, per = 0.0, {}
    cur = None
    for line in open(path):
        m = re.match(r"^(\S.*) dispatches (\d+)$", line.rstrip())
        if m:
            cur = None if any(s in m.group(1) for s in SKIP) else m.group(1)
            continue
        m = re.match(r"^\s+(\S+)\s+total (\S+)", line)
        if cur and m and m.group(1) == counter:
            per[cur] = per.get(cur, 0.0) + float(m.group(2))
            tot += float(m.group(2))
    return tot, per


f, fper = total(f"gpurun_out/{tag}_inc_{prec}_fetch_pmc.txt", "FETCH_SIZE")
w, wper = total(f"gpurun_out/{tag}_inc_{prec}_write_pmc.txt", "WRITE_SIZE")
runs_total = runs + 1     # inc_block.py runs the block once more to allocate its buffers
out = {"block": "inc double-conv forward (3->64->64 @320x427, train mode, batch 32)", "precision": prec,
       "hbm_bytes_per_run": (2 * f + w) * 1024.0 / runs_total,
       "read_bytes_per_run": 2 * f * 1024.0 / runs_total, "write_bytes_per_run": w * 1024.0 / runs_total,
       "per_kernel_KB": {k: {"FETCH_SIZE_x2": 2 * fper.get(k, 0.0) / runs_total, "WRITE_SIZE": wper.get(k, 0.0) / runs_total}
                         for k in sorted(set(fper) | set(wper))},
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> read bytes = 2 x FETCH_SIZE",
       "sources": [f"profiles/{tag}_inc_{prec}_pmc_fetch_size.txt", f"profiles/{tag}_inc_{prec}_pmc_write_size.txt"],
       "library_source_digest": (open("gelslim_depth_amd/csrc/libgsd.so.stamp").read().strip()
                                 if __import__("os").path.exists("gelslim_depth_amd/csrc/libgsd.so.stamp") else None)}
try:
    allj = json.load(open("profiles/inc_traffic.json"))
except (OSError, ValueError):
    allj = {}
allj[prec] = out
json.dump(allj, open("profiles/inc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
