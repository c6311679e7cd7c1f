#!/bin/bash
# the 12-bit form of kernel G1w for the o projection of 64-row windows (four column tiles per workgroup): parity, then end to end, same box
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
python -m pytest tests/test_gpu_glue.py -x -q -m gpu -k "g1z" 2>&1 | tail -2
B="python bench.py --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
run() { name=$1; shift
  env "$@" > $O/r6_wzo_$name.json 2> $O/r6_wzo_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/r6_wzo_$name.json").read().strip().splitlines()[-1])
    print("$name:", d["ms_per_step"], "ms/step   G1 avg", d["roofline"].get("avg_us"))
except Exception as e:
    print("$name FAILED", e); print(open("$O/r6_wzo_$name.err").read()[-1000:])
PY
}
for rep in 1 2; do
run emu3_bf16_on SJD_G1WZ=1 $B --model emu3_8b --dtype bf16 --window 32
run emu3_bf16_off SJD_G1WZ=0 $B --model emu3_8b --dtype bf16 --window 32
run lumina_2p_on SJD_G1WZ=1 SJD_G1_CFG='{"o":[512,4,1]}' $B --prompts-per-gpu 2
run lumina_2p_off SJD_G1WZ=0 $B --prompts-per-gpu 2
done
