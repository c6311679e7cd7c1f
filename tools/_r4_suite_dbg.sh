#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 600 python -X faulthandler -m pytest tests/test_gpu_api.py -q -x -s 2>&1 | grep -v amdgpu.ids | grep -v "^  File" | head -40 | cut -c1-300 | tee $O/suite_dbg.txt
echo ---- again, alone
timeout 600 python -m pytest tests/test_gpu_api.py -q -x -k "flexar_solver_with_renew" 2>&1 | tail -3
