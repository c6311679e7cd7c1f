#!/bin/bash
# end-to-end A/B of G1w launch shapes at eight prompts per forward (one box, interleaved)
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image --no-other-configs --prompts-per-gpu 8"
run() { r=$(SJD_G1_CFG="$2" $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"); echo "$1 $r"; }
for rep in 1 2; do
run default '{}'
run o_1024_2 '{"o":[1024,2,1]}'
run o_896_4 '{"o":[896,4,1]}'
run down_2752_2 '{"down":[2752,2,1]}'
run down_1408_8 '{"down":[1408,8,1]}'
run qkv_1024_8 '{"qkv":[1024,8,1]}'
run qkv_832_8 '{"qkv":[832,8,1]}'
run qkv_2048_3 '{"qkv":[2048,3,1]}'
run gu_2048_6 '{"gate_up":[2048,6,1]}'
run gu_2048_4 '{"gate_up":[2048,4,1]}'
done
