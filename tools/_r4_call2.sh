#!/bin/bash
# round 4, GPU call 2: where a tile's time goes in the ring kernel (timing probes), phase stamps, and the D-split MHA kernel
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4
O=gpurun_out/r4
kb() { timeout 300 python tools/k1_bench.py --graph "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['avg_us'])"; }
{
echo "# Emu3 shape (GQA 32/8, window 32, 16 splits), pair us per layer in a hipGraph; ring R=4"
for kv in 4096 8192; do
  for rep in 1 2; do
    for v in product k1r_nocompute k1r_nodma k1r_nosoftmax k1r_nobarrier k1r_nodma_nosoftmax k1r_nodma_nobarrier; do
      lib=tools/_exp/$v/libsjd_hip.so; [ $v == product ] && lib=accelerating-t2i-ar-with-sjd_amd/libsjd_hip.so
      echo -n "$v kv=$kv "; SJD_HIP_LIB=$lib SJD_K1_RING_SLOTS=4 kb --heads 32 --kv-heads 8 --window 32 --n-split 16 --kv-len $kv
    done
  done
done
echo "# n_split sweep, ring R=4, kv 4096"
for ns in 8 16 32; do echo -n "ring ns=$ns "; SJD_K1_RING_SLOTS=4 kb --heads 32 --kv-heads 8 --window 32 --n-split $ns --kv-len 4096; done
} 2>&1 | tee $O/k1_ring_probes.txt
SJD_K1_RING_SLOTS=4 timeout 600 python tools/phase_trace.py --k1s 2>/dev/null | tee $O/k1_ring_phase.jsonl
{
echo "# MHA (Lumina shape): k1_partial + k1_combine (4 splits) vs k1_dsplit, pair us per layer in a hipGraph"
for kv in 64 448 1216 2368; do
  for rep in 1 2; do
    echo -n "partial+combine kv=$kv "; kb --n-split 4 --kv-len $kv
    echo -n "dsplit          kv=$kv "; SJD_K1_DSPLIT=1 kb --n-split 4 --kv-len $kv
  done
done
} 2>&1 | tee $O/k1_dsplit_ab.txt
SJD_K1_DSPLIT=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k1_k3_attention or device_side_kv_len or ignores_nan" 2>&1 | tail -5 | tee $O/k1_dsplit_tests.txt
