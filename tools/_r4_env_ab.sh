#!/bin/bash
# HIP runtime switches that touch the cost of a dependent launch inside a graph: ms_per_step of the headline under each
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
Q="--steps 192 --warmup 16 --no-whole-image --no-floor --no-torch-baseline --no-cpu-baseline --no-other-configs"
run() { echo "---- $1"; env $1 timeout 600 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"; }
{
run "X=0"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "AMD_OPT_FLUSH=0"
run "ROC_USE_FGS_KERNARG=0"
run "DEBUG_HIP_KERNARG_COPY_OPT=0"
run "GPU_MAX_HW_QUEUES=1"
run "X=1"
} 2>&1 | tee $O/env_ab.txt
