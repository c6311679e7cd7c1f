#!/bin/bash
# F2 with its norm parameters / rotary frequency / position requested in front of the partial planes (late round 6) against the round-5 schedule, same box
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
B="python bench.py --steps 128 --warmup 8 --no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
for rep in 1 2 3; do
for v in new old; do
if [ $v == old ]; then export SJD_HIP_LIB=tools/_exp/f2_old/libsjd_hip.so; else unset SJD_HIP_LIB; fi
$B > $O/r6_f2h_$v.json 2> $O/r6_f2h_$v.err
python - <<PY
import json
d = json.loads(open("$O/r6_f2h_$v.json").read().strip().splitlines()[-1])
print("$v F2:", d["ms_per_step"], "ms/step")
PY
done
done
unset SJD_HIP_LIB
for v in new old; do
if [ $v == old ]; then export SJD_HIP_LIB=tools/_exp/f2_old/libsjd_hip.so; else unset SJD_HIP_LIB; fi
$B --prompts-per-gpu 8 --steps 64 > $O/r6_f2h8_$v.json 2> $O/r6_f2h8_$v.err
python - <<PY
import json
d = json.loads(open("$O/r6_f2h8_$v.json").read().strip().splitlines()[-1])
print("$v F2, 8 prompts:", d["ms_per_step"], "ms/step")
PY
done
