#!/bin/bash
# K1 pair per layer in a hipGraph against the number of key splits: Emu3's shape (ring kernel) and Lumina's (k1_partial)
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
{
for kv in 1024 4096 8192; do for ns in 4 8 12 16 24 32; do
  echo -n "emu3 kv=$kv n_split=$ns  "; timeout 300 python tools/k1_bench.py --graph --heads 32 --kv-heads 8 --window 32 --kv-len $kv --n-split $ns 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['avg_us'], 'us', d['gbps'], 'GB/s')"
done; done
for kv in 1216 2368; do for ns in 2 3 4 6 8; do
  echo -n "lumina kv=$kv n_split=$ns  "; timeout 300 python tools/k1_bench.py --graph --kv-len $kv --n-split $ns 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['avg_us'], 'us', d['gbps'], 'GB/s')"
done; done
} 2>&1 | tee $O/k1_split_sweep.txt
