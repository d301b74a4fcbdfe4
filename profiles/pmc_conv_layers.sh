#!/bin/bash
# HBM-side traffic of conv3x3_w2d_kernel per LAYER SHAPE (forward, batch 32): FETCH_SIZE and WRITE_SIZE in separate passes
# (GPU box, repo root):   bash profiles/pmc_conv_layers.sh <tag>   -> gpurun_out/<tag>_conv_layers_pmc.txt
tag=$1
export TMPDIR=/tmp PYTHONPATH=$PWD
res=$PWD/gpurun_out/${tag}_conv_layers_pmc.txt
: > $res
IFS=";" read -ra SHAPES <<< "${CONV_SHAPES:-64 64 320 427;128 128 160 213;256 256 80 106;512 512 40 53;1024 1024 20 26;1024 512 40 53;128 64 320 427}"
for shape in "${SHAPES[@]}"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    out=$PWD/gpurun_out/pmc_cl
    rm -rf $out
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- python3 profiles/one_conv_w2d.py $shape 32 4 > gpurun_out/${tag}_cl.log 2>&1 || { echo "rocprofv3 failed for $shape $ctr" >> $res; continue; }
    f=$(find $out -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$shape" $ctr "$(grep '^done' gpurun_out/${tag}_cl.log)" <<'PY' >> $res
import csv, sys, collections
tot = 0.0; disp = set()
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv3x3_w2d_kernel' not in r['Kernel_Name'] or r['Counter_Name'] != sys.argv[3]: continue
    tot += float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
n = max(len(disp), 1)
print("ci co h w = %-18s %-10s per launch %10.1f MB over %d launches | %s" % (sys.argv[2], sys.argv[3], tot / n / 1e3 * 1.024, n, sys.argv[4]))
PY
  done
done
cat $res
