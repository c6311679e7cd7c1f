#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 2400 python tools/fuzz_loops.py --n ${1:-120} --seed ${2:-41} $3 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-300 | tee $O/r4_fuzz_summary${3:+_poisoned}.txt
