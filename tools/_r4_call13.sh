#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_glue.py -x -q -k "mlp_pair" 2>&1 | tail -6 | cut -c1-400 | tee $O/mlp_pair_tests.txt
timeout 600 python tools/mlp_pair_bench.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/mlp_pair_bench.txt
