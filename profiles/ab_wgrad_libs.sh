#!/bin/bash
# A/B of diagnostic builds of the dW kernel on the 17 layers (GPU box): bash profiles/ab_wgrad_libs.sh <suffix> <suffix> ...
# (profiles/ubench/libgsd_wg2d_<suffix>.so from profiles/build_diag_one.sh; "prod" = the product library)
export PYTHONPATH=.
for v in "$@"; do
  if [ "$v" = prod ]; then unset GSD_LIB_PATH; else export GSD_LIB_PATH=$PWD/profiles/ubench/libgsd_wg2d_$v.so; fi
  python profiles/bench_wgrad_engine_form.py 32 2>&1 | grep -v amdgpu > gpurun_out/ab_$v.txt || exit 1
  echo "$v: $(tail -1 gpurun_out/ab_$v.txt)"
done
