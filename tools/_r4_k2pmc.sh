#!/bin/bash
# K2 / K4 in a real decode under PMC: where do the waves' cycles go (instruction fetch vs memory vs issue)?
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
OUT=$O/r4_k2_pmc.jsonl; : > $OUT
rocprofv3 -L 2>/dev/null | grep -oiE "\b(SQC_[A-Z_]*ICACHE[A-Z_]*|SQ_IFETCH[A-Z_]*|SQ_WAIT_INST[A-Z_]*|SQ_INST_CYCLES[A-Z_]*|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_INSTS_SALU|SQ_INSTS_VALU|SQ_INSTS_SMEM|SQ_INSTS_BRANCH|SQC_ICACHE[A-Z_]*)\b" | sort -u | tr '\n' ' ' > $O/pmc_available.txt
cat $O/pmc_available.txt; echo
B="python bench.py --model ${1:-lumina7b} ${2} --steps 48 --warmup 8 --no-floor --no-torch-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_IFETCH_LEVEL" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_k2_$T -- $B > /dev/null 2>> $O/pmc_k2.err
  echo "# $C" >> $OUT
  python tools/pmc_summary.py $O/pmc_k2_$T k2_logits k4_verify >> $OUT
  rm -rf $O/pmc_k2_$T
done
cat $OUT | cut -c1-400
