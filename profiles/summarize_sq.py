#!/usr/bin/env python3
"""Per-kernel derived figures from a run_pmc.sh SQ pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_INSTS_MFMA SQ_INSTS_VALU): matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs),
vector instructions per MFMA, share of wave cycles spent parked (s_waitcnt / barrier).
usage: python profiles/summarize_sq.py profiles/r04_f_pmc_sq.txt [min share of GUI_ACTIVE, default 0.003]"""
import re
import sys

path = sys.argv[1]
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.003
rows, cur = [], None
for line in open(path):
    m = re.match(r"^(\S.*) dispatches (\d+)$", line.rstrip())
    if m:
        cur = {"name": m.group(1), "n": int(m.group(2))}
        rows.append(cur)
        continue
    m = re.match(r"^\s+(\S+)\s+total (\S+)", line)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(2))
tot = sum(r.get("GRBM_GUI_ACTIVE", 0.0) for r in rows)
print("%-74s %5s %9s %8s %9s %8s %8s" % ("kernel", "disp", "ms/launch*", "share", "mfma_busy", "valu/mfma", "parked"))
for r in sorted(rows, key=lambda r: -r.get("GRBM_GUI_ACTIVE", 0.0)):
    ga = r.get("GRBM_GUI_ACTIVE", 0.0)
    if ga < floor * tot:
        continue
    cyc = ga / 8.0 / r["n"]
    busy = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (ga / 8.0 * 1024.0) if ga else 0.0
    mf = r.get("SQ_INSTS_MFMA", 0.0)
    print("%-74s %5d %9.3f %8.4f %9.3f %8.2f %8.3f" % (r["name"][:74], r["n"], cyc / 2.4e6, ga / tot, busy,
                                                    r.get("SQ_INSTS_VALU", 0.0) / mf if mf else float("nan"),
                                                    r.get("SQ_WAIT_ANY", 0.0) / max(r.get("SQ_WAVE_CYCLES", 1.0), 1.0)))
print("* GRBM_GUI_ACTIVE / 8 / dispatches at a nominal 2.4 GHz (the counter pass serialises kernels; use the --stats file for times)")
