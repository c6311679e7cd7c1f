#!/bin/bash
# G1w for 65..128-row windows: parity tests, then launch shapes end to end (four and three prompts per forward), same box
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
python -m pytest tests/test_gpu_glue.py -x -q -m gpu -k "three_and_four_row_tiles" 2>&1 | tail -3
python -m pytest tests/test_gpu_loop.py -x -q -m gpu -k "three_and_four_prompts or real_shape_batch" 2>&1 | tail -3
B="python bench.py --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
run() {  # name prompts env...
  name=$1; pp=$2; shift 2
  env "$@" $B --prompts-per-gpu $pp > $O/r6_w128_${name}_${pp}p.json 2> $O/r6_w128_${name}_${pp}p.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/r6_w128_${name}_${pp}p.json").read().strip().splitlines()[-1])
    print("$pp prompts  $name:", d["value"], "tok/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print("$pp $name", "FAILED", e); print(open("$O/r6_w128_${name}_${pp}p.err").read()[-1200:])
PY
}
for pp in 4 3; do
run tiled8 $pp SJD_G1_WIDE_128=0
run g1w_same_shapes $pp SJD_G1_WIDE_128=1
run g1w_tuned_all $pp SJD_G1_CFG='{"qkv":[832,8,1],"o":[512,4,1],"gate_up":[2048,6,1],"down":[1408,4,1]}'
run g1w_tuned_no_qkv $pp SJD_G1_CFG='{"o":[512,4,1],"gate_up":[2048,6,1],"down":[1408,4,1]}'
run g1w_tuned_gu_down $pp SJD_G1_CFG='{"gate_up":[2048,6,1],"down":[1408,4,1]}'
run tiled8_again $pp SJD_G1_WIDE_128=0
done
