#!/usr/bin/env python3
"""profiles/<tag>_pmc_summary.jsonl + profiles/<tag>_bench_kernel_stats.csv (tools/profile_round.sh) -> profiles/g1z_traffic.json, k1_traffic.json,
rocprof_g1.json: the figures bench.py's `roofline.traffic` and `roofline.rocprof_avg_us` quote, regenerated from the round's OWN passes
(the traffic file used to be a round-3 pass).  usage: make_traffic_json.py <tag> [profiles dir]"""
import csv
import json
import os
import sys

tag = sys.argv[1]
pdir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
rows, sect = {}, None
for line in open(os.path.join(pdir, f"{tag}_pmc_summary.jsonl")):
    line = line.strip()
    if line.startswith("#"):
        sect = line[1:].strip().rsplit("/", 1)[-1]
        continue
    if not line:
        continue
    r = json.loads(line)
    rows.setdefault(r["kernel"], {}).update({k: v for k, v in r.items() if k not in ("kernel", "dispatches")})
    rows[r["kernel"]]["dispatches"] = r["dispatches"]


def pick(sub):
    hit = [k for k in rows if sub in k and "FETCH_SIZE" in rows[k] and "WRITE_SIZE" in rows[k]]
    return (hit[0], rows[hit[0]]) if hit else (None, None)


corr = "x2: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream (MI355X_MICROARCH.md, HBM section)"
src = f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only (tools/profile_round.sh {tag} -> profiles/{tag}_pmc_summary.jsonl)"
# ---- G1z: the four projection launches of a layer, average per launch (three g1z_skinny_gemm shapes + one g1z_gateup_silu)
kz, z = pick("g1z_skinny_gemm<1")
ks, s = pick("g1z_gateup_silu<")
if z and s:
    rd = (3 * z["FETCH_SIZE"] + s["FETCH_SIZE"]) / 4 * 1024 * 2
    wr = (3 * z["WRITE_SIZE"] + s["WRITE_SIZE"]) / 4 * 1024
    alg, stored = 101560320, 76646400
    out = {"kernel": "g1z_skinny_gemm (q|k|v, o, down) + g1z_gateup_silu over the 12-bit lossless weight stream: the four projection launches of a layer, average per launch",
           "workload": "tools/g1z_bench.py --launches 48 (the product launch shapes, M = 32 rows, 8 weight copies so that every launch streams from HBM)",
           "source": src, "fetch_correction": corr, "g1z_skinny_gemm_FETCH_KB_raw": z["FETCH_SIZE"], "g1z_skinny_gemm_WRITE_KB_raw": z["WRITE_SIZE"],
           "g1z_gateup_silu_FETCH_KB_raw": s["FETCH_SIZE"], "g1z_gateup_silu_WRITE_KB_raw": s["WRITE_SIZE"],
           "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
           "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 3), "stored_bytes_per_launch": stored,
           "traffic_over_stored_bytes": round((rd + wr) / stored, 3),
           "definitions": "algorithmic = SURVEY.md 8(d): the bf16 weight matrix N*K*2 + the activation rows, what bench.py's roofline.achieved prices; "
                          "stored = the 12-bit stream + unit headers + activation rows, what the kernel has to move"}
    json.dump(out, open(os.path.join(pdir, "g1z_traffic.json"), "w"), indent=2)
    print("g1z_traffic.json", out["hbm_bytes_per_launch"], out["traffic_over_algorithmic"])
kk, k = pick("k1_partial<")
if k:
    rd, wr, alg = k["FETCH_SIZE"] * 1024 * 2, k["WRITE_SIZE"] * 1024, 39600128
    out = {"kernel": "k1_partial<bf16, D=128>", "workload": "tools/k1_bench.py --kv-len 1216 --n-split 4 --launches 96 --graph (B=2, H=H_kv=32, D=128, window 16)",
           "source": src, "FETCH_SIZE_avg_KB_raw": k["FETCH_SIZE"], "fetch_correction": corr, "WRITE_SIZE_avg_KB_raw": k["WRITE_SIZE"],
           "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "algorithmic_bytes_per_launch": alg,
           "hbm_bytes_per_launch": int(rd + wr), "traffic_over_algorithmic": round((rd + wr) / alg, 3)}
    json.dump(out, open(os.path.join(pdir, "k1_traffic.json"), "w"), indent=2)
    print("k1_traffic.json", out["hbm_bytes_per_launch"], out["traffic_over_algorithmic"])
# ---- Emu3's K1 (round 6): k1_partial_ring + k1_combine at GQA 32 / 8, window 32, kv 4186, 16 key splits -- the pair's traffic against its
# algorithmic bytes (2 B_cfg H_kv (kv + L) D e for K and V, once per kv head, + the q rows)
kr, r_ = pick("k1_partial_ring<")
kc, c_ = pick("k1_combine<")
if r_ and c_:
    kv, L_, Hkv, H, D_, B_ = 4186, 32, 8, 32, 128, 2
    alg = 2 * B_ * Hkv * (kv + L_) * D_ * 2 + B_ * L_ * H * D_ * 2
    rd = (r_["FETCH_SIZE"] + c_["FETCH_SIZE"]) * 1024 * 2
    wr = (r_["WRITE_SIZE"] + c_["WRITE_SIZE"]) * 1024
    out = {"kernel": "k1_partial_ring<bf16, D=128> + k1_combine (Emu3's shape)", "workload": "tools/k1_bench.py --kv-len 4186 --n-split 16 --heads 32 --kv-heads 8 --window 32 --launches 96 --graph",
           "source": src, "fetch_correction": corr, "partial_FETCH_KB_raw": r_["FETCH_SIZE"], "partial_WRITE_KB_raw": r_["WRITE_SIZE"],
           "combine_FETCH_KB_raw": c_["FETCH_SIZE"], "combine_WRITE_KB_raw": c_["WRITE_SIZE"], "hbm_read_bytes_per_pair": int(rd), "hbm_write_bytes_per_pair": int(wr),
           "hbm_bytes_per_pair": int(rd + wr), "algorithmic_bytes_per_pair": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 3)}
    json.dump(out, open(os.path.join(pdir, "k1_emu3_traffic.json"), "w"), indent=2)
    print("k1_emu3_traffic.json", out["hbm_bytes_per_pair"], out["traffic_over_algorithmic"])
# ---- rocprofv3 --kernel-trace --stats of the bench decode: average duration of a G1 launch (the dominant kernel)
stats = os.path.join(pdir, f"{tag}_bench_kernel_stats.csv")
if os.path.exists(stats):
    calls = tot = 0
    per = {}
    for r in csv.DictReader(open(stats)):
        n = r["Name"]
        if "g1z_skinny_gemm<1" in n or "g1z_gateup_silu<" in n or "g1_skinny_gemm<" in n or "g1_gateup_silu<" in n:
            calls += int(r["Calls"])
            tot += int(r["TotalDurationNs"])
            per[n.split("(")[0].replace("void ", "")] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)}
    if calls:
        out = {"source": f"profiles/{tag}_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py --steps 64, tools/profile_round.sh {tag})",
               "g1_launches": calls, "avg_us_per_g1_launch": round(tot / calls / 1e3, 2), "by_kernel": per,
               "note": "all G1 launches of the profiled decode, the output head's included (one in 129)"}
        json.dump(out, open(os.path.join(pdir, "rocprof_g1.json"), "w"), indent=2)
        print("rocprof_g1.json", out["avg_us_per_g1_launch"])
