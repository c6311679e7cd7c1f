#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4
O=gpurun_out/r4
kb() { timeout 300 python tools/k1_bench.py --graph "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['avg_us'])"; }
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k1" 2>&1 | tail -4 | tee $O/k1_tests_valu.txt
{
echo "# ring kernel after the VALU cuts (exp2 fold, packed cvt, conditional rescale); Emu3 shape, pair us per layer"
for kv in 1024 4096 8192; do for rep in 1 2; do
  echo -n "shared kv=$kv "; SJD_K1_RING=0 kb --heads 32 --kv-heads 8 --window 32 --n-split 16 --kv-len $kv
  for R in 4 6; do echo -n "ring R=$R kv=$kv "; SJD_K1_RING_SLOTS=$R kb --heads 32 --kv-heads 8 --window 32 --n-split 16 --kv-len $kv; done
done; done
} 2>&1 | tee $O/k1_ring_ab2.txt
SJD_K1_RING_SLOTS=4 timeout 600 python tools/phase_trace.py --k1s 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/k1_ring_phase.jsonl
timeout 1200 python -m pytest "tests/test_gpu_real_shape_forward.py" -x -q -k "emu3_8b_bf16 or 2300 or 4100" 2>&1 | tail -6 | tee $O/new_tests_b2.txt
cp gpurun_out/r4_real_shape_forward_*.json $O/ 2>/dev/null
