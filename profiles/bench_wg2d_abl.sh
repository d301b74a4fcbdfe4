#!/bin/bash
# dW 2-D Winograd kernel, ablation builds (profiles/build_diag_one.sh gsd_wgrad_w2d.hip "-DWG2D_ABL=<mask>" wg2d_abl<mask>) against
# the product on four layers: usage (GPU box): bash profiles/bench_wg2d_abl.sh "1 2 4 8 16 6 14"
export PYTHONPATH=.
LAYERS=inc.c1,down0.c1,down2.c1,up3.c0
echo "== product"; python profiles/bench_wgrad_engine_form.py 32 $LAYERS | grep -v amdgpu
for a in $1; do
  echo "== WG2D_ABL=$a"
  GSD_LIB_PATH=$PWD/profiles/ubench/libgsd_wg2d_abl$a.so python profiles/bench_wgrad_engine_form.py 32 $LAYERS | grep -v amdgpu
done
