#!/bin/bash
# G1w timing probes on one box: the product build and -DGW_NO_* builds under tools/_exp/gw_*.  usage: _gw_probe.sh SHAPE "CANDS" "VARIANTS" probe...
cd ${GRAFT_REPO_ROOT:-.}
SHAPE=${1:-qkv}; CAND=${2:-832:8:1,2048:4:1}; VARS=${3:-0}; shift 3
O=gpurun_out/g1w_probes.txt
: > $O
echo "## product" >> $O
python tools/g1w_bench.py --only $SHAPE --cand $CAND --variants $VARS --no-blas --no-old >> $O 2>&1
for v in "$@"; do
  echo "## $v" >> $O
  SJD_HIP_EXP_LIB=tools/_exp/gw_$v/libsjd_hip_exp.so python tools/g1w_bench.py --only $SHAPE --cand $CAND --variants $VARS --no-blas --no-old >> $O 2>&1
done
grep -v "best\|amdgpu.ids" $O | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('#'): print(l.strip()); continue
    try: r = json.loads(l)
    except Exception: print(l.strip()); continue
    print(r.get('kernel'), r.get('KC'), r.get('tiles'), 'v', r.get('variant'), r.get('us'))
"
