#!/usr/bin/env python3
"""What the per-unit escape of the 12-bit stream costs (round 6, csrc/sjd_gemm_raw.h): G1z / G1sz at the product launch shapes, 32 rows, over a
clean pack and over the same weights with a zero row planted in ~`--frac` of the (k-chunk, tile) units (those units become RAW and the fix-up
launch recomputes their tiles).  hipGraph replays over several weight copies.  One JSON line per shape."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sjd_amd.backbones as BB  # noqa: E402
import sjd_amd.ops as ops  # noqa: E402

SHAPES = dict(qkv=(12288, 4096), o=(4096, 4096), gate_up=(22016, 4096), down=(4096, 11008))


def timed_graph(fn, n):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    res = []
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    return min(res) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=32)
    ap.add_argument("--frac", type=float, default=0.01)
    ap.add_argument("--launches", type=int, default=24)
    ap.add_argument("--copies", type=int, default=6)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = BB.ChameleonBackbone.G1_CFG_Z
    for name, (N, K) in SHAPES.items():
        if a.only and name not in a.only.split(","):
            continue
        KC, waves, sm = cfg[name]
        gateup = name == "gate_up"
        g = torch.Generator().manual_seed(N + K)
        x = torch.randn(a.rows, K, generator=g).to(torch.bfloat16).to(dev)
        clean, dirty, stats = [], [], None
        for c in range(a.copies):
            w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
            clean.append(ops.pack_weight_z(w, KC, sm, gateup=gateup))
            n_chunks, T = (K + KC - 1) // KC, N // 32
            n_hit = max(1, int(round(a.frac * n_chunks * T)))
            pick = torch.randperm(n_chunks * T, generator=g)[:n_hit]
            for u in pick.tolist():                       # one zero row inside unit (chunk, tile)
                ci, t = divmod(u, T)
                w[32 * t + 7, ci * KC:min(K, ci * KC + KC)] = 0.0
            dirty.append(ops.pack_weight_z(w, KC, sm, gateup=gateup))
            stats = dirty[-1].stats
        if gateup:
            run = lambda ws: (lambda i: ops.gateup_silu(x, ws[i % a.copies], N // 2, K, sm))
        else:
            run = lambda ws: (lambda i: ops.skinny_gemm(x, ws[i % a.copies], N, K, KC, waves, sm))
        t_clean, t_dirty = timed_graph(run(clean), a.launches), timed_graph(run(dirty), a.launches)
        print(json.dumps(dict(shape=name, rows=a.rows, KC=KC, waves=waves, units=stats["units"], raw_units=stats["raw_units"],
                              raw_pairs=dirty[-1].n_raw_pairs, raw_frac=round(stats["raw_units"] / stats["units"], 4), clean_us=round(t_clean, 2),
                              with_raw_units_us=round(t_dirty, 2), slowdown=round(t_dirty / t_clean, 4))), flush=True)


if __name__ == "__main__":
    main()
