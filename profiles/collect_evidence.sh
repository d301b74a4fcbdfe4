#!/bin/bash
# Copy the summaries of an evidence run (bash profiles/r06_profile.sh <tag> all; python bench.py > gpurun_out/<tag>_bench_default.json)
# from gpurun_out/ into profiles/ under the names profiles/traffic*.json cite:   bash profiles/collect_evidence.sh <tag>
tag=${1:-r06_z}
set -e
for leg in fp32 fp32_b8 fp32_one_stream fp32_b8_one_stream bf16; do
  for f in stats.txt kernel_stats.csv bench.json; do
    [ -f gpurun_out/${tag}_${leg}_$f ] && cp gpurun_out/${tag}_${leg}_$f profiles/
  done
done
cp gpurun_out/${tag}_fetch_pmc.txt profiles/${tag}_pmc_fetch_size.txt
cp gpurun_out/${tag}_write_pmc.txt profiles/${tag}_pmc_write_size.txt
cp gpurun_out/${tag}_sq_pmc.txt profiles/${tag}_pmc_sq.txt
cp gpurun_out/${tag}_pmc_sq_summary.txt profiles/
for prec in bf16 fp32; do
  cp gpurun_out/${tag}_inc_${prec}_fetch_pmc.txt profiles/${tag}_inc_${prec}_pmc_fetch_size.txt
  cp gpurun_out/${tag}_inc_${prec}_write_pmc.txt profiles/${tag}_inc_${prec}_pmc_write_size.txt
done
cp gpurun_out/${tag}_inc_bf16_time.txt profiles/
cp gpurun_out/${tag}_traffic.json profiles/traffic.json
cp gpurun_out/${tag}_traffic_wgrad.json profiles/traffic_wgrad.json
cp gpurun_out/${tag}_inc_traffic.json profiles/inc_traffic.json
[ -f gpurun_out/${tag}_bench_default.json ] && cp gpurun_out/${tag}_bench_default.json profiles/
git status --short profiles | head -40
