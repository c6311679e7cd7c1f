#!/bin/bash
# Round profile set (run on the GPU box through gpurun; outputs under gpurun_out/prof_*; copy the summaries into profiles/):
#   1. rocprofv3 --kernel-trace --stats of a short bench run -> per-kernel time, split by launch shape
#   2. PMC passes (counters only, no trace domains besides kernel-trace): MFMA busy for K1 and G1, HBM traffic for K1 / G1 / K2+head
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=${1:-r6}
PREV=${2:-profiles/r5final_bench_by_shape.txt}      # the previous round's by-shape table: the regression guard compares against it
O=gpurun_out
mkdir -p $O
B="python bench.py --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-ar-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_trace -- $B > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof_bench.err
python tools/trace_by_grid.py $O/prof_${TAG}_trace 200 > $O/${TAG}_bench_by_shape.txt
find $O/prof_${TAG}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_bench_kernel_stats.csv
K1="python tools/k1_bench.py --kv-len 1216 --n-split 4 --launches 96 --graph"
# (round 6, VERDICT r5 #5: the Emu3 shape too -- GQA 32 / 8, window 32, kv 4186, 16 key splits: k1_partial_ring + k1_combine -- so that the "2.07 x the
#  algorithmic bytes" figure comes from the current tree, not from round 4)
K1E="python tools/k1_bench.py --kv-len 4186 --n-split 16 --heads 32 --kv-heads 8 --window 32 --launches 96 --graph"
G1="python tools/g1z_bench.py --launches 48"      # G1 / G1s and G1z / G1sz at the product launch shapes (kernel names g1_* / g1z_*)
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/prof_${TAG}_k1_$T -- $K1 > /dev/null 2>> $O/${TAG}_pmc.err
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/prof_${TAG}_g1_$T -- $G1 > /dev/null 2>> $O/${TAG}_pmc.err
done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/prof_${TAG}_k1e_$C -- $K1E > /dev/null 2>> $O/${TAG}_pmc.err
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/prof_${TAG}_k1_$C -- $K1 > /dev/null 2>> $O/${TAG}_pmc.err
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/prof_${TAG}_g1_$C -- $G1 > /dev/null 2>> $O/${TAG}_pmc.err
done
: > $O/${TAG}_pmc_summary.jsonl
for d in $O/prof_${TAG}_k1_* $O/prof_${TAG}_k1e_* $O/prof_${TAG}_g1_*; do
  echo "# $d" >> $O/${TAG}_pmc_summary.jsonl
  python tools/pmc_summary.py $d k1_ g1_ g1z_ >> $O/${TAG}_pmc_summary.jsonl
done
# the figures bench.py quotes (roofline.traffic, roofline.rocprof_avg_us) from THIS round's passes: copy the three files into profiles/
python tools/make_traffic_json.py ${TAG} $O
cat $O/${TAG}_pmc_summary.jsonl
head -30 $O/${TAG}_bench_by_shape.txt
# keep the merged output small: the raw trace directories stay on the box
rm -rf $O/prof_${TAG}_*
# per-kernel regression guard (round 6): any hot kernel more than 3 % slower than in the previous round's committed table fails the script
if [ -f "$PREV" ]; then python tools/regression_guard.py $PREV $O/${TAG}_bench_by_shape.txt | tee $O/${TAG}_regression_guard.txt; GUARD=${PIPESTATUS[0]}; else GUARD=0; fi
exit $GUARD
