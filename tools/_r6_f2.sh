#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
python -m pytest tests/test_gpu_glue.py -x -q -m gpu -k "f2" 2>&1 | tail -5
python -m pytest tests/test_gpu_loop.py -x -q -m gpu -k "five_to_eight or three_and_four" 2>&1 | tail -3
B="python bench.py --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
for pp in 8 4; do
for fl in 1 0; do
SJD_F2_ROWS=$fl $B --prompts-per-gpu $pp > $O/r6_f2_${pp}p_$fl.json 2> $O/r6_f2_${pp}p_$fl.err
python - <<PY
import json
try:
    d = json.loads(open("$O/r6_f2_${pp}p_$fl.json").read().strip().splitlines()[-1])
    print("$pp prompts  F2 rows kernel $fl:", d["value"], "tok/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print("$pp", "FAILED", e); print(open("$O/r6_f2_${pp}p_$fl.err").read()[-1500:])
PY
done
done
