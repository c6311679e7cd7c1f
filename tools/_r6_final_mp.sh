#!/bin/bash
# round-end record: the multi-prompt bench lines on ONE box (2, 3, 4, 5, 6, 8 prompts per forward) + the 128-row G1w sweep the DESIGN cites
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
B="python bench.py --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
for pp in 8 6 5 4 3 2; do
$B --prompts-per-gpu $pp > $O/r6_final_bench_${pp}prompts.json 2> $O/r6_final_bench_${pp}prompts.err
python - <<PY
import json
try:
    d = json.loads(open("$O/r6_final_bench_${pp}prompts.json").read().strip().splitlines()[-1])
    print("$pp prompts", d["value"], "tok/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print("$pp", "FAILED", e); print(open("$O/r6_final_bench_${pp}prompts.err").read()[-1500:])
PY
done
python tools/g1w_bench.py --rows 128 --variants 0,20 --no-blas > $O/r6_g1w_sweep_128rows.jsonl 2> $O/r6_g1w_sweep_128rows.err; tail -8 $O/r6_g1w_sweep_128rows.jsonl
