#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: every launch between the last two adam_ema kernels in start order
(offset from the step's start, duration, queue), then the per-kernel sums of that step by queue.
  python3 profiles/step_timeline.py <..._kernel_trace.csv> [--sums-only]"""
import collections
import csv
import sys


def kname(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    d = 0
    for i, c in enumerate(n):
        if c == '<':
            d += 1
        elif c == '>':
            d -= 1
        elif c == '(' and d == 0:
            return n[:i]
    return n


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_ema' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['End_Timestamp'])
step = rows[a + 1:b + 1]
print(f"step: {(int(rows[b]['End_Timestamp']) - t0) / 1e6:.3f} ms, {len(step)} launches")
if '--sums-only' not in sys.argv:
    for r in step:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(f"{(s - t0) / 1e6:8.3f} {(e - s) / 1e3:8.1f} us q{r['Queue_Id']} grid {r['Grid_Size_X']:>9s} wg {r['Workgroup_Size_X']:>4s} {kname(r['Kernel_Name'])[:60]}")
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    k = (r['Queue_Id'], kname(r['Kernel_Name']))
    agg[k][0] += 1
    agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for q in sorted({k[0] for k in agg}):
    tot = sum(v[1] for k, v in agg.items() if k[0] == q)
    print(f"queue {q}: busy {tot / 1e6:.3f} ms")
    for k, v in sorted(agg.items(), key=lambda x: -x[1][1]):
        if k[0] == q:
            print(f"   {k[1][:60]:60s} {v[0]:4d} launches {v[1] / 1e6:8.3f} ms")
