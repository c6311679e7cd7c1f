#!/bin/bash
# Lumina output head: split-K chunk / waves per workgroup of the G1 launch against K2's plane count (K2 sums n_chunks planes per column, one CU per row)
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
Q="--steps 128 --warmup 16 --no-whole-image --no-floor --no-torch-baseline --no-cpu-baseline --no-other-configs"
for cfg in "[1024,4,true]" "[2048,4,true]" "[2048,2,true]" "[2048,8,true]" "[4096,2,true]" "[4096,4,true]"; do
  echo "---- HEAD_CFG $cfg"
  SJD_HEAD_CFG="$cfg" rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hc -- python bench.py $Q > $O/hc.json 2>/dev/null
  python tools/trace_by_grid.py $O/prof_hc 200 | grep "k2_logits\|k4_verify\|(8[0-9][0-9][0-9], [0-9], 1)\|(4[0-9][0-9][0-9], [0-9], 1)\|(16[0-9][0-9][0-9], [0-9], 1)" | cut -c1-150
  python -c "import json; d=json.loads(open('$O/hc.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
  rm -rf $O/prof_hc
done 2>&1 | tee $O/head_cfg_sweep.txt
