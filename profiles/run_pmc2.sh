#!/bin/bash
# usage (GPU box, repo root): bash profiles/run_pmc2.sh <tag> "<counters>" <kernel-name-substring> python3 <script> [args]
# One rocprofv3 --pmc pass over an arbitrary command; prints per-dispatch averages of the kernels whose name contains the substring.
tag=$1; shift; ctrs=$1; shift; kern=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- "$@" > gpurun_out/${tag}_pmc.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" "$kern" <<'PY' | tee gpurun_out/${tag}_pmc.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:60]
    if sys.argv[2] not in k:
        continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'], k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k, n in cnt.most_common(4):
    print(k, "dispatches", n)
    for c, v in sorted(agg[k].items()):
        print("    %-36s per-dispatch %.5g" % (c, v / n))
PY
