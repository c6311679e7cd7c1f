#!/bin/bash
# kernel resource usage of one translation unit, one line per kernel matching $2:  tools/kres.sh sjd_gemm g1_wide
cd "$(dirname "$0")/../accelerating-t2i-ar-with-sjd_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16 $KRES_FLAGS -Rpass-analysis=kernel-resource-usage --cuda-device-only -S -o ${KRES_OUT:-/tmp/kres.s} $1.hip 2>&1 |
  awk -v pat="$2" '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name); on = (name ~ pat)}
       on && /(TotalSGPRs|VGPRs:|AGPRs|Spill|Occupancy|LDS Size)/ {v=$0; sub(/.*remark: +/,"",v); sub(/ \[-Rpass.*/,"",v); line = line " | " v}
       on && /LDS Size/ {print name line; line=""}'
