#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k2 or k4" 2>&1 | tail -4 | cut -c1-300 | tee $O/k2k4_spread_tests.txt
timeout 900 python tools/phase_trace.py --in-situ 2>&1 | grep -v amdgpu.ids | grep "k2_" | cut -c1-700 | tee $O/k2_phase.jsonl
bash tools/_r4_prof.sh r4c lumina7b emu3_8b 2>&1 | grep -E "k2_|k4_|ms_per_step"
