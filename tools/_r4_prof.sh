#!/bin/bash
# tools/_r4_prof.sh TAG [models...]: by-shape kernel times (rocprofv3 --kernel-trace --stats) of short decodes -> gpurun_out/r4/TAG_<model>_by_shape.txt
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
TAG=$1; shift
for M in "$@"; do
  case $M in
    lumina7b) fl="";;
    emu3_8b) fl="--dtype bf16 --window 32";;
    emu3_8b_fp16) fl="--window 32";;
    anole7b) fl="";;
  esac
  name=${M%_fp16}
  B="python bench.py --model $name $fl --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$M -- $B > $O/${TAG}_${M}_bench.json 2> $O/${TAG}_${M}.err
  python tools/trace_by_grid.py $O/prof_$M 200 > $O/${TAG}_${M}_by_shape.txt
  find $O/prof_$M -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_${M}_kernel_stats.csv
  head -16 $O/${TAG}_${M}_by_shape.txt | cut -c1-150
  python -c "import json,sys; d=json.loads(open('$O/${TAG}_${M}_bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'tok/step', d['tokens_per_step'])"
  rm -rf $O/prof_$M
done
