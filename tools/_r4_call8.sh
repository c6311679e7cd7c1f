#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
kb() { timeout 300 python tools/k1_bench.py --graph "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['avg_us'])"; }
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k1" 2>&1 | tail -3 | tee $O/k1_tests_rs.txt
{
echo "# k1_combine with one (SJD_K1_COMBINE_RS=1) / two workgroups per (batch, head, chunk): pair us per layer"
for rep in 1 2; do
  for kv in 4096; do
    echo -n "emu3 rs1 kv=$kv "; SJD_K1_COMBINE_RS=1 SJD_K1_RING_SLOTS=4 kb --heads 32 --kv-heads 8 --window 32 --n-split 16 --kv-len $kv
    echo -n "emu3 rs2 kv=$kv "; SJD_K1_RING_SLOTS=4 kb --heads 32 --kv-heads 8 --window 32 --n-split 16 --kv-len $kv
  done
  for kv in 448 1216 2368; do
    echo -n "lumina rs1 kv=$kv "; SJD_K1_COMBINE_RS=1 kb --n-split 4 --kv-len $kv
    echo -n "lumina rs2 kv=$kv "; kb --n-split 4 --kv-len $kv
  done
done
} 2>&1 | tee $O/k1_combine_rs.txt
# by-shape kernel times of the two decodes (K2 / K4 after the bisection select + LDS staging)
for M in "lumina7b:" "emu3_8b:--dtype bf16 --window 32"; do
  name=${M%%:*}; fl=${M#*:}
  B="python bench.py --model $name $fl --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-cpu-baseline --no-whole-image --no-other-configs"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- $B > $O/prof_${name}_bench.json 2> $O/prof_${name}.err
  python tools/trace_by_grid.py $O/prof_$name 200 > $O/r4a_${name}_by_shape.txt
  head -22 $O/r4a_${name}_by_shape.txt
  rm -rf $O/prof_$name
done
