#!/bin/bash
# usage: bash profiles/run_pmc_script.sh <tag> "<counters>" <script.py> [args]   -- one PMC pass over a python script, per-kernel totals
tag=$1; shift; ctrs=$1; shift
export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python3 "$@" > gpurun_out/${tag}_pmc.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' > gpurun_out/${tag}_pmc.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'], k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k, n in cnt.most_common(20):
    print(k, "dispatches", n)
    for c, v in sorted(agg[k].items()):
        print("    %-32s total %.4g  per-dispatch %.4g" % (c, v, v / n))
PY
cat gpurun_out/${tag}_pmc.txt
