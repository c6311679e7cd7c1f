#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
kb() { timeout 300 python tools/k1_bench.py --graph "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['avg_us'])"; }
SJD_K1_RING_HALVES=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "k1_k3_attention and (emu3 or ragged or gqa)" 2>&1 | tail -3 | tee $O/k1_ring_halves_tests.txt
{
echo "# Emu3 shape, pair us per layer: one 8-wave workgroup per (batch, kv head, split) [ring R=4 / R=6] vs two 4-wave workgroups (each its own ring, R=4)"
for kv in 1024 4096 8192; do for rep in 1 2; do
  echo -n "8 waves R=4 kv=$kv "; SJD_K1_RING_SLOTS=4 kb --heads 32 --kv-heads 8 --window 32 --n-split 16 --kv-len $kv
  echo -n "8 waves R=6 kv=$kv "; kb --heads 32 --kv-heads 8 --window 32 --n-split 16 --kv-len $kv
  echo -n "2 x 4 waves kv=$kv "; SJD_K1_RING_HALVES=1 kb --heads 32 --kv-heads 8 --window 32 --n-split 16 --kv-len $kv
  echo -n "2 x 4 waves, 8 splits kv=$kv "; SJD_K1_RING_HALVES=1 kb --heads 32 --kv-heads 8 --window 32 --n-split 8 --kv-len $kv
done; done
} 2>&1 | tee $O/k1_ring_halves.txt
