#!/bin/bash
# round 4 PMC passes (counters only + kernel-trace; one counter set per run) for K1: Emu3 shape ring vs round-3 shared-tile kernel, Lumina / Anole
# shapes key split vs column split.  Output: gpurun_out/r4/r4_k1_pmc.jsonl
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
OUT=$O/r4_k1_pmc.jsonl; : > $OUT
run() {   # tag, env assignments, k1_bench args
  local tag=$1; shift; local envs=$1; shift
  for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    T=$(echo $C | tr ' ' '_' | cut -c1-40)
    env $envs rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${tag}_$T -- python tools/k1_bench.py --graph --launches 96 "$@" > /dev/null 2>> $O/pmc.err
    echo "# $tag: $C" >> $OUT
    python tools/pmc_summary.py $O/pmc_${tag}_$T k1_ >> $OUT
    rm -rf $O/pmc_${tag}_$T
  done
}
run emu3_ring   "SJD_K1_RING=1" --heads 32 --kv-heads 8 --window 32 --n-split 16 --kv-len 4176
run emu3_shared "SJD_K1_RING=0" --heads 32 --kv-heads 8 --window 32 --n-split 16 --kv-len 4176
run lumina_keysplit "SJD_X=0" --n-split 4 --kv-len 1216
run lumina_colsplit "SJD_X=0" --colsplit --kv-len 448
run lumina_keysplit_448 "SJD_X=0" --n-split 4 --kv-len 448
run anole_colsplit "SJD_X=0" --fp8 --colsplit --kv-len 594
run anole_keysplit "SJD_X=0" --fp8 --n-split 1 --kv-len 594
wc -l $OUT
