#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k2" 2>&1 | tail -6 | cut -c1-400 | tee $O/k2_zero_state_tests.txt
timeout 1800 python -m pytest tests/test_gpu_loop.py tests/test_gpu_golden_loop.py tests/test_gpu_api.py -x -q 2>&1 | tail -4 | cut -c1-300 | tee $O/loops_zero_state.txt
bash tools/_r4_prof.sh r4d lumina7b emu3_8b 2>&1 | grep -E "k2_|k4_|ms_per_step"
