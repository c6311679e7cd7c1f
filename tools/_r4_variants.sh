#!/bin/bash
# the K1 tests and the real-shape forward checks on the kernel variants the defaults do not pick, allocator poisoned
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
t() { echo "---- [$1] $2"; env SJD_TEST_POISON=1 $1 timeout 1500 python -m pytest $2 -q -x 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200; }
{
t "SJD_K1_RING=0" "tests/test_gpu_kernels.py -k k1"
t "SJD_K1_RING=0" "tests/test_gpu_real_shape_forward.py -k emu3"
t "SJD_K1_REGIME=keysplit" "tests/test_gpu_kernels.py -k k1"
t "SJD_K1_REGIME=keysplit" "tests/test_gpu_loop.py"
t "SJD_K1_RING_SLOTS=6" "tests/test_gpu_real_shape_forward.py -k emu3"
} 2>&1 | tee $O/variants_poisoned.txt
