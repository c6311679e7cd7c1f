#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k2 or k4" 2>&1 | tail -4 | cut -c1-300 | tee $O/k2k4_bisect_tests.txt
bash tools/_r4_prof.sh r4b lumina7b emu3_8b
