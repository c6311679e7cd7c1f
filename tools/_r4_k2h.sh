#!/bin/bash
# K2 with 0 / 1 / 3 / 5 / 7 helper workgroups per row (SJD_K2_HELPERS): kernel time by shape, Lumina and Emu3
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
for H in 0 1 3 5 7; do
  export SJD_K2_HELPERS=$H
  echo "---- SJD_K2_HELPERS=$H"
  bash tools/_r4_prof.sh r4k2h$H lumina7b emu3_8b 2>&1 | grep "k2_logits\|k4_verify\|ms_per_step" | cut -c1-150
done 2>&1 | tee $O/k2_helpers_ab.txt
