#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | cut -c1-300 | tee $O/full_gpu_suite.txt
