#!/bin/bash
# round 6: K5 / K2 / K4 of every slot in one launch each (sjd_*_slots) -- parity tests, then the same-box A/B of the multi-prompt step time
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
python -m pytest tests/test_gpu_loop.py -x -q -m gpu -k "prompts or batch or continuous or per_slot" 2>&1 | tail -5
B="python bench.py --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
for pp in 8 4 2; do
for sl in 1 0; do
SJD_SLOT_LAUNCHES=$sl $B --prompts-per-gpu $pp > $O/r6_slots_${pp}p_$sl.json 2> $O/r6_slots_${pp}p_$sl.err
python - <<PY
import json
try:
    d = json.loads(open("$O/r6_slots_${pp}p_$sl.json").read().strip().splitlines()[-1])
    print("$pp prompts  slot launches $sl:", d["value"], "tok/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print("$pp", "FAILED", e); print(open("$O/r6_slots_${pp}p_$sl.err").read()[-1500:])
PY
done
done
