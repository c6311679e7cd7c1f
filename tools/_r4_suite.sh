#!/bin/bash
# the whole GPU suite twice: as the driver runs it (-x), then with the allocator's free pool poisoned before every test (tests/conftest.py)
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 3000 python -X faulthandler -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300 | tee $O/full_gpu_suite_run1.txt
SJD_TEST_POISON=1 timeout 3000 python -X faulthandler -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300 | tee $O/full_gpu_suite_poisoned.txt
