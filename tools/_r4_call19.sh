#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
kb() { timeout 300 python tools/k1_bench.py --graph "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['avg_us'])"; }
SJD_K1_DSPLIT_PF=9 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "k1_k3_attention and colsplit" 2>&1 | tail -6 | cut -c1-300 | tee $O/k1_dsplit_ring_tests.txt
{
echo "# MHA (Lumina shape), pair us per layer in a hipGraph: key split (4) + combine | column split (fragment loads, one tile ahead) | column split over an LDS-DMA ring of full rows"
for kv in 64 448 1216 2368; do for rep in 1 2; do
  echo -n "keysplit+combine kv=$kv "; kb --n-split 4 --kv-len $kv
  echo -n "colsplit         kv=$kv "; kb --colsplit --kv-len $kv
  echo -n "colsplit ring    kv=$kv "; SJD_K1_DSPLIT_PF=9 kb --colsplit --kv-len $kv
done; done
} 2>&1 | tee $O/k1_dsplit_ring_ab.txt
