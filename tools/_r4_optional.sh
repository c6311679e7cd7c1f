#!/bin/bash
# the optional (off by default) fused paths, allocator poisoned: whole-loop parity and the real-shape forward checks
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
t() { echo "---- [$1] $2"; env SJD_TEST_POISON=1 $1 timeout 1500 python -m pytest $2 -q -x 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200; }
{
t "SJD_MLP_PAIR=1" "tests/test_gpu_loop.py tests/test_gpu_real_shape_forward.py -k lumina"
t "SJD_K1_FUSED=1" "tests/test_gpu_loop.py tests/test_gpu_real_shape_forward.py -k lumina"
t "SJD_REDUCE_FUSED=1 SJD_G1Z=0" "tests/test_gpu_loop.py tests/test_gpu_real_shape_forward.py -k lumina"
t "SJD_K1_MERGED=1" "tests/test_gpu_loop.py tests/test_gpu_real_shape_forward.py -k lumina"
} 2>&1 | tee $O/optional_paths_poisoned.txt
