#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_philox.py -x -q -k "k2 or k4 or philox" 2>&1 | tail -6 | cut -c1-400 | tee $O/k2k4_compact_tests.txt
timeout 1800 python -m pytest tests/test_gpu_loop.py tests/test_gpu_golden_loop.py tests/test_gpu_api.py -x -q 2>&1 | tail -4 | cut -c1-300 | tee $O/loops_compact.txt
bash tools/_r4_prof.sh r4e lumina7b emu3_8b 2>&1 | grep -E "k2_|k4_|ms_per_step"
timeout 900 python tools/phase_trace.py --in-situ 2>&1 | grep -v amdgpu.ids | grep "k2_" | cut -c1-700 | tee $O/k2_phase_compact.jsonl
