#!/usr/bin/env python3
"""G1s (gate|up projection with SiLU * up as its epilogue) alone: hipGraph replays over several packed weight copies (every launch streams
from HBM), HIP-event time per launch; the driver for the PMC passes behind profiles/g1s_traffic.json.
  python tools/g1s_bench.py [--launches 48] [--copies 6] [--unfused]      (--unfused: G1 + F3 on the same weights, for comparison)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sjd_amd.ops as ops

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=48)
ap.add_argument("--copies", type=int, default=6)
ap.add_argument("--unfused", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
inter, K = 11008, 4096
x = torch.randn(32, K, device=dev).to(torch.bfloat16)
wps = [ops.pack_weight((torch.randn(2 * inter, K, device=dev) / K ** 0.5).to(torch.bfloat16), K // 2, True) for _ in range(a.copies)]
rn = (ops.residual_sumsq(x.clone(), None), K, 1e-5)


def one(i):
    if a.unfused:
        return ops.silu_mul(ops.skinny_gemm(x, wps[i % a.copies], 2 * inter, K, K // 2, waves=8, step_major=True), rows=32, dtype=x.dtype, row_norm=rn)
    return ops.gateup_silu(x, wps[i % a.copies], inter, K, True, row_norm=rn)


with torch.cuda.stream(torch.cuda.Stream()):
    one(0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(a.launches):
        one(i)
g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4):
    g.replay()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (4 * a.launches)
alg = 2 * inter * K * 2 + 32 * K * 2 + 32 * inter * 2
print(json.dumps(dict(kernel="g1_skinny_gemm + f3_silu_mul" if a.unfused else "g1_gateup_silu", us_per_launch=round(us, 2), algorithmic_bytes=alg,
                      TBps=round(alg / us / 1e6, 3))))
