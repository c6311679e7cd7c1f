#!/bin/bash
# end-to-end A/B of the 12-bit stream's launch shapes for o and down (Lumina, one box): bench ms per step, two runs each, interleaved
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
Q="--steps 192 --warmup 16 --no-whole-image --no-floor --no-torch-baseline --no-cpu-baseline --no-other-configs"
run() { echo -n "$1  "; env SJD_G1_CFG="$2" python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
{
for i in 1 2; do
run "default (o 512/6/tile-major, down 768/8/tile-major)" ""
run "o 512/4/step-major                                 " '{"o": [512, 4, true]}'
run "down 896/8/tile-major                              " '{"down": [896, 8, false]}'
run "both                                               " '{"o": [512, 4, true], "down": [896, 8, false]}'
done
} 2>&1 | tee $O/g1cfg_ab.txt
