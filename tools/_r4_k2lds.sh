#!/bin/bash
# K2 / K4 at Emu3's shape (rows staged in LDS): LDS activity and bank conflicts
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
OUT=$O/r4_k2_lds_pmc.jsonl; : > $OUT
B="python bench.py --model emu3_8b --dtype bf16 --window 32 --steps 32 --warmup 8 --no-floor --no-torch-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_k2l_$T -- $B > /dev/null 2>> $O/pmc_k2l.err
  echo "# $C" >> $OUT
  python tools/pmc_summary.py $O/pmc_k2l_$T k2_logits k4_verify >> $OUT
  rm -rf $O/pmc_k2l_$T
done
cat $OUT | cut -c1-400
