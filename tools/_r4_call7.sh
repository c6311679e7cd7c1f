#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
SJD_K1_RING_SLOTS=4 timeout 600 python tools/phase_trace.py --k1s > $O/pt.out 2> $O/pt.err; echo rc=$?
grep -v amdgpu.ids $O/pt.err | tail -30 | cut -c1-500; tail -5 $O/pt.out | cut -c1-600
