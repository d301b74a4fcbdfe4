#!/bin/bash
# Round-6 evidence run (GPU box, repo root): rocprofv3 kernel stats of the bench command (fp32 batch 32 = the metric, fp32 batch 8 = the
# per-GPU share of configs[3] on 8 GPUs, bf16 batch 32), the three PMC passes behind profiles/traffic.json (dominant kernel:
# conv3x3_w2d_kernel; every kernel of the step listed), and the two PMC passes over the bf16 `inc` block behind
# profiles/inc_traffic.json.  Results land in gpurun_out/ (copy the summaries into profiles/ afterwards).
#   bash profiles/r05_profile.sh [tag] [part: all | stats | pmc | inc]
tag=${1:-r06_z}
part=${2:-all}
export PYTHONPATH=$GRAFT_REPO_ROOT
if [ "$part" = all ] || [ "$part" = stats ]; then
  bash profiles/run_prof.sh ${tag}_fp32 --no-extra --steps 5 --warmup 2 > gpurun_out/${tag}_fp32_prof.log 2>&1
  echo "fp32 stats done"; head -8 gpurun_out/${tag}_fp32_stats.txt
  bash profiles/run_prof.sh ${tag}_fp32_b8 --no-extra --batch 8 --steps 5 --warmup 2 > gpurun_out/${tag}_fp32_b8_prof.log 2>&1
  # the same two with the weight gradients on the main stream (GSD_SIDE_DW=0): kernels one after the other, so that a kernel's
  # duration is its own (with the side stream a small launch of the chain can wait 0.3-0.8 ms for a CU and its duration says so)
  GSD_SIDE_DW=0 bash profiles/run_prof.sh ${tag}_fp32_one_stream --no-extra --steps 5 --warmup 2 > gpurun_out/${tag}_fp32_one_stream_prof.log 2>&1
  GSD_SIDE_DW=0 bash profiles/run_prof.sh ${tag}_fp32_b8_one_stream --no-extra --batch 8 --steps 5 --warmup 2 > gpurun_out/${tag}_fp32_b8_one_stream_prof.log 2>&1
  echo "fp32 batch-8 stats done"; head -8 gpurun_out/${tag}_fp32_b8_stats.txt
  bash profiles/run_prof.sh ${tag}_bf16 --no-extra --dtype bf16 --steps 5 --warmup 2 > gpurun_out/${tag}_bf16_prof.log 2>&1
  echo "bf16 stats done"; head -6 gpurun_out/${tag}_bf16_stats.txt
fi
if [ "$part" = all ] || [ "$part" = pmc ]; then
  bash profiles/run_pmc.sh ${tag}_fetch "FETCH_SIZE" --no-extra --steps 1 --warmup 0 > /dev/null 2>&1; echo fetch done
  bash profiles/run_pmc.sh ${tag}_write "WRITE_SIZE" --no-extra --steps 1 --warmup 0 > /dev/null 2>&1; echo write done
  bash profiles/run_pmc.sh ${tag}_sq "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" --no-extra --steps 1 --warmup 0 > /dev/null 2>&1; echo sq done
  python3 profiles/make_traffic.py ${tag} conv3x3_w2d_kernel
  cp profiles/traffic.json gpurun_out/${tag}_traffic.json
  python3 profiles/make_traffic.py ${tag} wgrad3x3_w2d_kernel profiles/traffic_wgrad.json   # the second kernel of the step
  cp profiles/traffic_wgrad.json gpurun_out/${tag}_traffic_wgrad.json
  python3 profiles/summarize_sq.py gpurun_out/${tag}_sq_pmc.txt > gpurun_out/${tag}_pmc_sq_summary.txt
fi
if [ "$part" = all ] || [ "$part" = inc ]; then
  bash profiles/run_pmc_script.sh ${tag}_inc_bf16_fetch "FETCH_SIZE" profiles/inc_block.py bf16 32 3 > /dev/null 2>&1; echo inc fetch done
  bash profiles/run_pmc_script.sh ${tag}_inc_bf16_write "WRITE_SIZE" profiles/inc_block.py bf16 32 3 > /dev/null 2>&1; echo inc write done
  python3 profiles/make_inc_traffic.py ${tag} bf16 3
  bash profiles/run_pmc_script.sh ${tag}_inc_fp32_fetch "FETCH_SIZE" profiles/inc_block.py fp32 32 3 > /dev/null 2>&1; echo inc fp32 fetch done
  bash profiles/run_pmc_script.sh ${tag}_inc_fp32_write "WRITE_SIZE" profiles/inc_block.py fp32 32 3 > /dev/null 2>&1; echo inc fp32 write done
  python3 profiles/make_inc_traffic.py ${tag} fp32 3
  cp profiles/inc_traffic.json gpurun_out/${tag}_inc_traffic.json
  python3 profiles/inc_block.py bf16 32 10 > gpurun_out/${tag}_inc_bf16_time.txt 2>&1; cat gpurun_out/${tag}_inc_bf16_time.txt
fi
