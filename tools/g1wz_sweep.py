#!/usr/bin/env python3
"""Launch-shape sweep of kernel G1w's 12-bit form (EXPERIMENTAL library: sjd_skinny_gemm_z_wide, 2 / 3 / 4 / 6 / 8 column tiles per workgroup, late round 6) at a given
row count, hipGraph replays over several weight copies; --product times the product's 12-bit kernels (sjd_skinny_gemm_z, <= 128 rows) on the same points.
  python tools/g1wz_sweep.py --rows 64 [--emu3] [--only qkv,down]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sjd_amd._lib as L  # noqa: E402
import sjd_amd.ops as ops  # noqa: E402
from g1z_bench import timed_graph  # noqa: E402

KCS = dict(qkv=(512, 832, 1024, 2048), o=(512, 1024, 2048), gate_up=(1024, 2048), down_l=(768, 896, 1408, 2752), down_e=(896, 1024, 1792, 2048))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--emu3", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--copies", type=int, default=3)
    ap.add_argument("--launches", type=int, default=24)
    ap.add_argument("--points", default="", help="shape:KC:tiles[,...] instead of the grid")
    ap.add_argument("--product", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load()
    shapes = dict(qkv=(12288, 4096), o=(4096, 4096), gate_up=(22016, 4096), down=(4096, 11008))
    if a.emu3:
        shapes = dict(qkv=(6144, 4096), o=(4096, 4096), gate_up=(28672, 4096), down=(4096, 14336))
    pts = {}
    if a.points:
        for p in a.points.split(","):
            s_, kc, t = p.split(":")
            pts.setdefault(s_, []).append((int(kc), int(t)))
    for name, (N, K) in shapes.items():
        if a.only and name not in a.only.split(","):
            continue
        if pts and name not in pts:
            continue
        x = torch.randn(a.rows, K, device=dev).to(torch.bfloat16)
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(a.copies)]
        rows = []
        grid = pts.get(name) or [(kc, t) for kc in KCS["down_e" if (name == "down" and a.emu3) else "down_l" if name == "down" else name] for t in (2, 3, 4, 6, 8)]
        for KC in sorted({kc for kc, _ in grid}):
            wzs = [ops.pack_weight_z(w, KC, True) for w in ws]
            for kc, tiles in grid:
                if kc != KC or (a.rows > 64 and a.rows <= 96 and tiles == 2):
                    continue
                if a.product:
                    avg, _ = timed_graph(lambda i: ops.skinny_gemm(x, wzs[i % a.copies], N, K, KC, tiles, True), a.launches, lib)
                else:
                    avg, _ = timed_graph(lambda i: ops.skinny_gemm_z_wide(x, wzs[i % a.copies], tiles), a.launches, lib)
                r = dict(shape=name, rows=a.rows, KC=KC, tiles=tiles, planes=-(-K // KC), workgroups=-(-(N // 32) // tiles) * -(-K // KC), us=round(avg * 1e3, 2),
                         kernel="product G1z" if a.product else "G1w 12-bit (experimental)")
                rows.append(r)
                print(json.dumps(r), flush=True)
            del wzs
            torch.cuda.empty_cache()
        print(json.dumps(dict(shape=name, best=sorted(rows, key=lambda r: r["us"])[:4])), flush=True)


if __name__ == "__main__":
    main()
