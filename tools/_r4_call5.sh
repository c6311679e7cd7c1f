#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4
O=gpurun_out/r4
df -h /tmp . | tail -3
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k2 or k4" 2>&1 | tail -25 | cut -c1-400 | tee $O/k2k4_bisect_tests.txt
