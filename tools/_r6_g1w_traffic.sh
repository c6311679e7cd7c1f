#!/bin/bash
# HBM traffic of kernel G1w from its OWN PMC passes (the multi-prompt bench lines quoted a round-3 file measured on another kernel):
# the four projections at the product launch shapes, 256 and 128 rows, FETCH_SIZE and WRITE_SIZE in separate passes with --kernel-trace only
# -> gpurun_out/g1w_traffic.json (copy to profiles/: bench.py quotes it in roofline.traffic for 65..256-row uncompressed decodes)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out
: > $O/g1w_traffic_raw.jsonl
for ROWS in 256 128; do
  if [ $ROWS == 256 ]; then CFG="qkv=2048:4:1 o=1024:2:1 gate_up=2048:8:1 down=1408:4:1"; else CFG="qkv=2048:4:1 o=896:4:1 gate_up=2048:6:1 down=1408:4:1"; fi
  for SC in $CFG; do
    S=${SC%%=*}; C=${SC##*=}
    for CT in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $CT --kernel-trace --output-format csv -d $O/prof_gwt -- python tools/g1w_bench.py --rows $ROWS --only $S --cand $C --variants 0 --no-blas --no-old --launches 6 --copies 6 > /dev/null 2>> $O/g1w_traffic.err
      python tools/pmc_summary.py $O/prof_gwt g1_wide | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); r.update(rows=$ROWS, shape='$S', cand='$C'); print(json.dumps(r))" >> $O/g1w_traffic_raw.jsonl
      rm -rf $O/prof_gwt
    done
  done
done
python - <<'PY'
import json
SH = dict(qkv=(12288, 4096), o=(4096, 4096), gate_up=(22016, 4096), down=(4096, 11008))
rows = [json.loads(l) for l in open("gpurun_out/g1w_traffic_raw.jsonl")]
out = {"kernel": "g1_wide (kernel G1w, csrc/sjd_gemm_wide.h): the four projections of a Lumina-7B layer at the product launch shapes, average per launch",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only (tools/_r6_g1w_traffic.sh, round 6, tools/g1w_bench.py over six weight copies)",
       "fetch_correction": "x2: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream (MI355X_MICROARCH.md, HBM section)"}
for R in (256, 128):
    per = {}
    for r in rows:
        if r["rows"] != R:
            continue
        d = per.setdefault(r["shape"], {"cand": r["cand"]})
        for k in ("FETCH_SIZE", "WRITE_SIZE"):
            if k in r:
                d[k + "_KB_raw"] = r[k]
        d["avg_us_under_pmc"] = r.get("avg_us_under_pmc")
    tot = alg = 0
    for s, d in per.items():
        N, K = SH[s]
        d["hbm_bytes"] = int(2 * d["FETCH_SIZE_KB_raw"] * 1024 + d["WRITE_SIZE_KB_raw"] * 1024)
        d["algorithmic_bytes"] = N * K * 2 + R * K * 2
        tot += d["hbm_bytes"]; alg += d["algorithmic_bytes"]
    out[f"rows_{R}"] = {"per_shape": per, "hbm_bytes_per_launch": tot // max(len(per), 1), "algorithmic_bytes_per_launch": alg // max(len(per), 1),
                        "traffic_over_algorithmic": round(tot / max(alg, 1), 3),
                        "note": "traffic above the algorithmic bytes = the fp32 split-K planes the launch writes (the consumers read them back); the activation re-reads are L2-resident"}
json.dump(out, open("gpurun_out/g1w_traffic.json", "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "per_shape"}) for k, v in out.items()}, indent=1))
PY
