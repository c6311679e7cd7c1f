#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
B="python bench.py --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
for pp in 8 6 5 4 3; do
$B --prompts-per-gpu $pp > $O/r6_${pp}p.json 2> $O/r6_${pp}p.err
python - <<PY
import json
try:
    d = json.loads(open("$O/r6_${pp}p.json").read().strip().splitlines()[-1])
    print("$pp prompts", d["value"], d["ms_per_step"], d["roofline"].get("avg_us"), d["roofline"].get("frac"))
except Exception as e:
    print("$pp", "FAILED", e); print(open("$O/r6_${pp}p.err").read()[-1500:])
PY
done
python -m pytest tests/test_gpu_loop.py -x -q -m gpu -k "five_to_eight or three_and_four or batch or continuous" 2>&1 | tail -5
