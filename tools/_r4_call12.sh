#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
kb() { timeout 300 python tools/k1_bench.py --graph "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['avg_us'])"; }
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k1_k3" 2>&1 | tail -4 | cut -c1-300 | tee $O/k1_colsplit_tests.txt
{
echo "# fp8 KV cache (Anole shape: MHA 32 heads, window 16): key split (auto: 1 split below 160 KB per head, else 4) + combine vs column split; pair us per layer"
for kv in 64 320 594 1040 1600 2368; do for rep in 1 2; do
  echo -n "fp8 keysplit ns=1 kv=$kv "; kb --fp8 --n-split 1 --kv-len $kv
  echo -n "fp8 keysplit ns=4 kv=$kv "; kb --fp8 --n-split 4 --kv-len $kv
  echo -n "fp8 colsplit      kv=$kv "; kb --fp8 --colsplit --kv-len $kv
done; done
} 2>&1 | tee $O/k1_dsplit_fp8_ab.txt
for M in "lumina7b:" "anole7b:"; do
  name=${M%%:*}
  for R in keysplit auto; do
    if [ $R == auto ]; then unset SJD_K1_REGIME; else export SJD_K1_REGIME=$R; fi
    python bench.py --model $name --steps 64 --warmup 8 --no-floor --no-torch-baseline --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', '$R', 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'whole', d.get('whole_image',{}).get('steady_ms_per_step'), d.get('per_kv'))"
  done
done 2>&1 | tee $O/regime_e2e.txt
