#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
for i in 1 2; do
timeout 3000 python -X faulthandler -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300 | tee $O/full_gpu_suite_run$i.txt
done
