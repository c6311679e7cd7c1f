#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout 900 python -m pytest "tests/test_gpu_real_shape_forward.py" -x -q -k "4100" 2>&1 | grep -E "AssertionError|assert |first|hip16|Error" | cut -c1-900 | head -20 | tee $O/emu3_4100_ring.txt
SJD_K1_RING=0 timeout 900 python -m pytest "tests/test_gpu_real_shape_forward.py" -x -q -k "4100" 2>&1 | grep -E "AssertionError|assert |first|hip16|Error|passed|failed" | cut -c1-900 | head -20 | tee $O/emu3_4100_shared.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "k2 or k4" 2>&1 | tail -6 | tee $O/k2k4_bisect_tests.txt
SJD_K1_RING_SLOTS=4 timeout 600 python tools/phase_trace.py --k1s 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/k1_ring_phase.jsonl
