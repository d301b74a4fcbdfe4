#!/bin/bash
# ablation of conv3x3_w2d_kernel (diagnostic libraries from profiles/build_diag_one.sh gsd_conv3x3_w2d.hip -DW2D_ABL=<bits>): what the
# weight fills, the halo fills, the wait for them, the MFMAs and the operand transform cost over the level 0-2 layer set at batch 32
for a in prod 1 2 3 4 8 16; do
  if [ "$a" = prod ]; then unset GSD_LIB_PATH; else export GSD_LIB_PATH=$PWD/profiles/ubench/libgsd_w2d_abl$a.so; fi
  echo "== W2D_ABL=$a (waves ${GSD_W2D_WAVES:-8})"
  python profiles/bench_conv_w2d.py 32 2 2>&1 | grep -E "^total|K256  160x213|M64   K64 "
done
