#!/bin/bash
# XCD-aware (column group, K chunk) map of kernel G1w: parity, per-launch A/B at the product shapes, end to end
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
python -m pytest tests/test_gpu_glue.py -x -q -m gpu -k "five_to_eight_row or three_and_four_row or two_row_tiles or chunk_neighbours" 2>&1 | tail -3
for X in 1 0 1 0; do
  echo "== SJD_G1W_XMAP=$X  256 rows"
  for SC in qkv=2048:4:1 o=1024:2:1 gate_up=2048:8:1 down=1408:4:1; do
    SJD_G1W_XMAP=$X python tools/g1w_bench.py --rows 256 --only ${SC%%=*} --cand ${SC##*=} --no-blas --no-old 2>/dev/null | grep '"kernel": "wide"' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('   ', r['shape'], r['KC'], r['tiles'], r['us'])"
  done
done
for X in 1 0; do
  echo "== SJD_G1W_XMAP=$X  128 rows"
  for SC in qkv=2048:4:1 o=896:4:1 gate_up=2048:6:1 down=1408:4:1 o=512:4:1 o=1024:4:1; do
    SJD_G1W_XMAP=$X python tools/g1w_bench.py --rows 128 --only ${SC%%=*} --cand ${SC##*=} --no-blas --no-old 2>/dev/null | grep '"kernel": "wide"' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('   ', r['shape'], r['KC'], r['tiles'], r['us'])"
  done
done
B="python bench.py --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
for pp in 8 4; do
for X in 1 0 1 0; do
SJD_G1W_XMAP=$X $B --prompts-per-gpu $pp > $O/r6_xmap_${pp}p_$X.json 2> $O/r6_xmap_${pp}p_$X.err
python - <<PY
import json
try:
    d = json.loads(open("$O/r6_xmap_${pp}p_$X.json").read().strip().splitlines()[-1])
    print("$pp prompts  XMAP=$X:", d["ms_per_step"], "ms/step", d["value"], "tok/s")
except Exception as e:
    print("$pp", "FAILED", e); print(open("$O/r6_xmap_${pp}p_$X.err").read()[-1200:])
PY
done
done
