#!/bin/bash
# G1w for 33..64-row windows on the uncompressed stream: Emu3-8B in fp16 (BASELINE config 3 as worded), end to end, same box
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
B="python bench.py --model emu3_8b --window 32 --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
T='{"qkv":[1024,4,1],"o":[512,4,1],"gate_up":[2048,8,1],"down":[1792,4,1]}'
run() {
  name=$1; shift
  env "$@" $B > $O/r6_w64_$name.json 2> $O/r6_w64_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/r6_w64_$name.json").read().strip().splitlines()[-1])
    print("$name:", d["ms_per_step"], "ms/step", d["value"], "tok/s   G1 avg", d["roofline"].get("avg_us"), "us frac", d["roofline"].get("frac"))
except Exception as e:
    print("$name", "FAILED", e); print(open("$O/r6_w64_$name.err").read()[-1200:])
PY
}
run r5_kernels SJD_G1_WIDE_64=0
run g1w_old_shapes SJD_G1_WIDE_64=1
run g1w_tuned_fused SJD_G1_CFG="$T"
run g1w_tuned_unfused SJD_G1_CFG="$T" SJD_GATEUP_FUSED=0
run g1w_old_shapes_unfused SJD_GATEUP_FUSED=0
run r5_kernels_again SJD_G1_WIDE_64=0
