#!/bin/bash
# Round-3 evidence run (GPU box, repo root): rocprofv3 kernel stats of the bench command + the three PMC passes behind
# profiles/traffic.json.  Results land in gpurun_out/ (copy the summaries into profiles/ afterwards).
tag=${1:-r03_z}
bash profiles/run_prof.sh ${tag}_fp32 --no-extra --steps 5 --warmup 2 > gpurun_out/${tag}_fp32_prof.log 2>&1
echo "fp32 stats done"; head -8 gpurun_out/${tag}_fp32_stats.txt
# (GSD_BF16_SIDE_DW=0: per-kernel durations without the side stream's co-runner; the step itself is timed with it by bench.py)
GSD_BF16_SIDE_DW=0 bash profiles/run_prof.sh ${tag}_bf16 --no-extra --dtype bf16 --steps 5 --warmup 2 > gpurun_out/${tag}_bf16_prof.log 2>&1
echo "bf16 stats done"; head -6 gpurun_out/${tag}_bf16_stats.txt
bash profiles/run_pmc.sh ${tag}_fetch "FETCH_SIZE" --no-extra --steps 1 --warmup 0 > /dev/null 2>&1; echo fetch done
bash profiles/run_pmc.sh ${tag}_write "WRITE_SIZE" --no-extra --steps 1 --warmup 0 > /dev/null 2>&1; echo write done
bash profiles/run_pmc.sh ${tag}_sq "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" --no-extra --steps 1 --warmup 0 > /dev/null 2>&1; echo sq done
python3 profiles/make_traffic.py ${tag}
cp profiles/traffic.json gpurun_out/${tag}_traffic.json
# the north-star block on its own (one-level network): real HBM traffic of the inc double-conv forward, both precisions
for prec in bf16 fp32; do
  bash profiles/run_pmc_script.sh ${tag}_inc_${prec}_fetch "FETCH_SIZE" profiles/inc_block.py $prec 32 1 > /dev/null 2>&1
  bash profiles/run_pmc_script.sh ${tag}_inc_${prec}_write "WRITE_SIZE" profiles/inc_block.py $prec 32 1 > /dev/null 2>&1
  python3 profiles/make_inc_traffic.py ${tag} $prec 1
  PYTHONPATH=. python3 profiles/inc_block.py $prec 32 10 | tee gpurun_out/${tag}_inc_${prec}_time.txt
done
cp profiles/inc_traffic.json gpurun_out/${tag}_inc_traffic.json
