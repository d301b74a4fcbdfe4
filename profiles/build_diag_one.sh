#!/bin/bash
# Diagnostic library (NOT the product) that differs from libgsd.so in ONE source compiled with extra flags:
#   bash profiles/build_diag_one.sh gsd_bf16_conv.hip "-DGCONV_ABL=4" abl4   ->  profiles/ubench/libgsd_abl4.so  (use with GSD_LIB_PATH)
# The other objects are the product build's (gelslim_depth_amd/csrc/obj; run `python -m gelslim_depth_amd.build` first).
set -e
cd "$(dirname "$0")/.."
src=gelslim_depth_amd/csrc
obj=$src/obj
mkdir -p profiles/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $2 -c $src/$1 -o profiles/ubench/diag_$3.o
others=$(ls $obj/*.o | grep -v "/${1%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o profiles/ubench/libgsd_$3.so profiles/ubench/diag_$3.o $others
echo built profiles/ubench/libgsd_$3.so
