#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 900 python tools/phase_trace.py --in-situ 2>&1 | grep -v amdgpu.ids | grep "k2_" | cut -c1-700 | tee $O/k2_phase.jsonl
