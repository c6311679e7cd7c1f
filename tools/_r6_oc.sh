#!/bin/bash
# which side leg of the default bench run perturbs the other_configs legs behind it?  (Emu3 bf16: 3.84 ms alone / after the headline only, 3.98 in the default run)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out
ALL="--no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image"
for keep in floor torch-baseline ar-baseline cpu-baseline whole-image; do
  FL=$(echo $ALL | sed "s/--no-$keep//")
  python bench.py --steps 64 --warmup 8 $FL > $O/r6_oc_$keep.json 2> $O/r6_oc_$keep.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/r6_oc_$keep.json").read().strip().splitlines()[-1])
    print("with $keep:", d["ms_per_step"], {k: (v.get("ms_per_step"), v["roofline"].get("avg_us")) for k, v in d.get("other_configs", {}).items()})
except Exception as e:
    print("$keep FAILED", e); print(open("$O/r6_oc_$keep.err").read()[-800:])
PY
done
