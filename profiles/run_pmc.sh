#!/bin/bash
# usage: bash profiles/run_pmc.sh <tag> "<counters>" [bench args]   -- one PMC pass, per-kernel averages
tag=$1; shift; ctrs=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-parity "$@" > gpurun_out/${tag}_pmc.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' > gpurun_out/${tag}_pmc.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
def kname(n):
    # kernel name with its template arguments, without return type and parameter list (a 60-character cut used to merge
    # or hide template variants)
    n = n.strip()
    if n.startswith('void '):
        n = n[5:]
    n = n.replace('(anonymous namespace)::', '')
    d = 0
    for i, ch in enumerate(n):
        d += ch == '<'
        d -= ch == '>'
        if ch == '(' and d == 0:
            n = n[:i]
            break
    return n[:160]
for r in csv.DictReader(open(sys.argv[1])):
    k = kname(r['Kernel_Name'])
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'], k)
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k, n in cnt.most_common():      # EVERY kernel of the run (the first twelve by dispatch count used to hide dW / ConvT / bf16 conv)
    print(k, "dispatches", n)
    for c, v in sorted(agg[k].items()):
        print("    %-32s total %.4g  per-dispatch %.4g" % (c, v, v / n))
PY
cat gpurun_out/${tag}_pmc.txt
