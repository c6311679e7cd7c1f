#!/bin/bash
# round 4, GPU call 1: LDS-DMA probe, K1 parity with the ring kernel, ring vs shared A/B at the Emu3 shape, the new tests, one bench line
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout 60 tools/lds_dma_probe > $O/lds_dma_probe.jsonl 2>&1; echo "probe rc=$?" | tee -a $O/log.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k1" 2>&1 | tail -15 | tee $O/k1_tests.txt
{
for kv in 1024 4096 8192; do
  for rep in 1 2; do
    echo -n "shared kv=$kv "; SJD_K1_RING=0 timeout 300 python tools/k1_bench.py --graph --heads 32 --kv-heads 8 --window 32 --n-split 16 --kv-len $kv | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['avg_us'])"
    for R in 4 6 8; do
      echo -n "ring R=$R kv=$kv "; SJD_K1_RING_SLOTS=$R timeout 300 python tools/k1_bench.py --graph --heads 32 --kv-heads 8 --window 32 --n-split 16 --kv-len $kv | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['avg_us'])"
    done
  done
done
# two / four prompts per forward at the Lumina shape (MHA, 2 / 4 chunks of 16 rows... pairs = chunks): window 32 / 64 rows per batch row
for kv in 1216; do
  for w in 64; do
    echo -n "shared mha window=$w kv=$kv "; SJD_K1_RING=0 timeout 300 python tools/k1_bench.py --graph --window $w --n-split 2 --kv-len $kv | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['avg_us'])"
    echo -n "ring   mha window=$w kv=$kv "; timeout 300 python tools/k1_bench.py --graph --window $w --n-split 2 --kv-len $kv | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['avg_us'])"
  done
done
} 2>&1 | tee $O/k1_ring_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parallel.py tests/test_gpu_loop.py::test_batch_engine_unseeded_slots_draw_independent_noise -x -q 2>&1 | tail -8 | tee $O/new_tests_a.txt
timeout 2400 python -m pytest tests/test_gpu_real_shape_forward.py -x -q 2>&1 | tail -8 | tee $O/new_tests_b.txt
timeout 1500 python -m pytest tests/test_gpu_glue.py -x -q -k "g1z_matches or g1sz_matches" 2>&1 | tail -5 | tee $O/new_tests_c.txt
timeout 900 python bench.py --steps 64 --warmup 8 > $O/bench_call1.json 2> $O/bench_call1.err; echo "bench rc=$?" | tee -a $O/log.txt
tail -c 600 $O/bench_call1.err
