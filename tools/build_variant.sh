#!/bin/bash
# tools/build_variant.sh NAME "FLAGS" [unit]: the libraries with csrc/<unit>.hip (default sjd_gemm) compiled under extra flags -> tools/_exp/NAME/
#   libsjd_hip.so      the product library with that unit rebuilt      (same-box A/B: SJD_HIP_LIB=tools/_exp/NAME/libsjd_hip.so python ...)
#   libsjd_hip_exp.so  the experimental library with that unit rebuilt (SJD_HIP_EXP_LIB=tools/_exp/NAME/libsjd_hip_exp.so: tools/g1w_bench.py, ...)
# Needs the product objects (make -C accelerating-t2i-ar-with-sjd_amd/csrc).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/accelerating-t2i-ar-with-sjd_amd/csrc
U=${3:-sjd_gemm}
mkdir -p $R/tools/_exp/$1
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16"
/opt/rocm/bin/hipcc $COMMON $2 -c $C/$U.hip -o $R/tools/_exp/$1/v.o &
/opt/rocm/bin/hipcc $COMMON -DSJD_EXPERIMENTAL $2 -c $C/$U.hip -o $R/tools/_exp/$1/v.exp.o &
wait
OBJS=""; OBJS_EXP=""
for o in sjd_sampling sjd_attention sjd_glue sjd_gemm sjd_capi; do
  if [ "$o" == "$U" ]; then OBJS="$OBJS $R/tools/_exp/$1/v.o"; OBJS_EXP="$OBJS_EXP $R/tools/_exp/$1/v.exp.o";
  else OBJS="$OBJS $C/$o.o"; if [ -f $C/$o.exp.o ]; then OBJS_EXP="$OBJS_EXP $C/$o.exp.o"; else OBJS_EXP="$OBJS_EXP $C/$o.o"; fi; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_exp/$1/libsjd_hip.so $OBJS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_exp/$1/libsjd_hip_exp.so $OBJS_EXP
rm -f $R/tools/_exp/$1/v.o $R/tools/_exp/$1/v.exp.o
