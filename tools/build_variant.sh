#!/bin/bash
# tools/build_variant.sh NAME "FLAGS": libsjd_hip.so with csrc/sjd_gemm.hip compiled under extra flags -> tools/_exp/NAME/libsjd_hip.so
# (same-box A/B runs: SJD_HIP_LIB=tools/_exp/NAME/libsjd_hip.so python tools/g1z_bench.py ...).  Needs the product library's objects (make).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/accelerating-t2i-ar-with-sjd_amd/csrc
mkdir -p $R/tools/_exp/$1
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16"
/opt/rocm/bin/hipcc $COMMON $2 -c $C/${3:-sjd_gemm}.hip -o $R/tools/_exp/$1/v.o
OBJS=""
for o in sjd_sampling sjd_attention sjd_glue sjd_gemm sjd_capi; do
  if [ "$o" == "${3:-sjd_gemm}" ]; then OBJS="$OBJS $R/tools/_exp/$1/v.o"; else OBJS="$OBJS $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_exp/$1/libsjd_hip.so $OBJS
rm -f $R/tools/_exp/$1/v.o
