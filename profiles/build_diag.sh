#!/bin/bash
# Diagnostic library (NOT the product): libgsd with in-kernel s_memtime stamps, for profiles/stamp_*.py on the GPU box.
#   bash profiles/build_diag.sh "-DGSD_W43_STAMPS"   ->  profiles/ubench/libgsd_diag.so   (use with GSD_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
src=gelslim_depth_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude $1 -o profiles/ubench/libgsd_diag.so \
  $src/gsd_conv3x3.hip $src/gsd_conv3x3_w43.hip $src/gsd_convT.hip $src/gsd_wgrad.hip $src/gsd_wgrad_w43.hip $src/gsd_wgrad_first.hip $src/gsd_pointwise.hip \
  $src/gsd_dataset.hip $src/gsd_bf16_conv.hip $src/gsd_bf16_pointwise.hip $src/gsd_bf16_wgrad.hip $src/gsd_bf16_first.hip
echo built profiles/ubench/libgsd_diag.so
