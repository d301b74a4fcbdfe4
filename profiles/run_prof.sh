#!/bin/bash
# usage (on the GPU box, from the repo root): bash profiles/run_prof.sh <tag> [bench args...]
# Runs bench.py under rocprofv3 --kernel-trace --stats and leaves a per-kernel summary in gpurun_out/<tag>_stats.txt
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-parity "$@" > gpurun_out/${tag}_bench.log 2>&1
grep '^{' gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench.json
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY' > gpurun_out/${tag}_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total GPU kernel time %.2f ms" % (tot / 1e6))
for r in rows[:22]:
    print("%-62s calls %5s total_ms %9.2f avg_us %9.1f pct %6.2f" % (r['Name'][:62], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
cat gpurun_out/${tag}_stats.txt
cut -c1-200 gpurun_out/${tag}_bench.json
