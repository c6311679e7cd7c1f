#!/bin/bash
# PMC passes over the wide kernel (q|k|v at 256 rows): LDS conflicts, MFMA busy, wait breakdown.  usage: _gw_pmc.sh "KC:tiles:sm" variant [lib]
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
C=${1:-832:8:1}; V=${2:-0}; SHAPE=${4:-qkv}
[ -n "$3" ] && export SJD_HIP_EXP_LIB=$3
O=gpurun_out
B="python tools/g1w_bench.py --only $SHAPE --cand $C --variants $V --no-blas --no-old --launches 6 --copies 3"
: > $O/gw_pmc.jsonl
for CT in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  T=$(echo $CT | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $CT --kernel-trace --output-format csv -d $O/prof_gw_$T -- $B > /dev/null 2>> $O/gw_pmc.err
  python tools/pmc_summary.py $O/prof_gw_$T g1_wide >> $O/gw_pmc.jsonl
  rm -rf $O/prof_gw_$T
done
cat $O/gw_pmc.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); r.pop('kernel'); print(r)"
