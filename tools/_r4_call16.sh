#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4; mkdir -p $O
timeout 900 python bench.py > $O/bench_mid.json 2> $O/bench_mid.err; echo "bench rc=$?"
timeout 900 python bench.py --prompts-per-gpu 4 --steps 64 --warmup 8 > $O/bench_4prompts.json 2> $O/bench_4p.err; echo "bench4 rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_mid.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','tokens_per_step','vs_baseline']})
print('whole', d['whole_image']); print('per_kv', d['per_kv'])
print('roofline', {k:d['roofline'][k] for k in ['achieved','frac','avg_us','rows']})
print('k1', {k:v for k,v in d['roofline_k1'].items() if k in ['kernel','achieved','frac','avg_us','pair_in_graph']})
for k,v in d['other_configs'].items(): print(k, v.get('ms_per_step'), v.get('tokens_per_s'), v.get('roofline',{}).get('frac'), v.get('roofline_k1',{}).get('pair_in_graph'), v.get('error'))
d=json.loads(open('gpurun_out/r4/bench_4prompts.json').read().strip().splitlines()[-1])
print('4 prompts', {k:d.get(k) for k in ['value','ms_per_step','tokens_per_step']})
PY
