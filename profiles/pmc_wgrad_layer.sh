#!/bin/bash
# PMC counters of the conv3x3 dW kernels on ONE layer of the engine-form harness (GPU box, repo root):
#   bash profiles/pmc_wgrad_layer.sh <tag> <layer> "<counters>"      -> gpurun_out/<tag>_pmc.txt (per-kernel, per-dispatch averages)
tag=$1; layer=$2; ctrs=$3
export TMPDIR=/tmp PYTHONPATH=$PWD
out=$PWD/gpurun_out/pmc_$tag
rm -rf $out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python3 profiles/bench_wgrad_engine_form.py 32 $layer > gpurun_out/${tag}_pmc.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' > gpurun_out/${tag}_pmc.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][:90]
    if 'wgrad' not in k: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'], k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k, n in cnt.most_common():
    print(k, "dispatches", n)
    for c, v in sorted(agg[k].items()):
        print("    %-32s per-dispatch %.5g" % (c, v / n))
PY
cat gpurun_out/${tag}_pmc.txt
