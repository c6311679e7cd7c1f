#!/usr/bin/env python3
"""Does the 256 MiB Infinity Cache keep a packed weight that a prefetch kernel has just read, and how fast does G1 stream it from
there?  Per projection shape (Lumina-7B), hipGraph-timed over 12 cycling weight copies (so every "cold" launch streams from HBM):
   cold   : N x G1(w_i)                       -- the round-1 operating point
   pf     : N x prefetch(w_i)                 -- the prefetch kernel's own rate, per grid size
   hot    : N x [prefetch(w_i); G1(w_i)] - pf -- G1 right after its weights were pulled through the cache
   overlap: N x [fork: prefetch(w_{i+1}) || filler(kernel that is latency-bound); join; G1(w_{i+1})]  (side stream, as in the engine)
Prints one JSON line per shape.  Measurement aid only."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sjd_amd._lib as L  # noqa: E402
import sjd_amd.ops as ops  # noqa: E402
import sjd_amd.backbones as BB  # noqa: E402

SHAPES = dict(qkv=(12288, 4096), o=(4096, 4096), gate_up=(22016, 4096), down=(4096, 11008))


def graph_time(body, n):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        body(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            body(i)
    g.replay()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(res)[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=12)
    ap.add_argument("--launches", type=int, default=48)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L.load()
    cfgs = BB.ChameleonBackbone.G1_CFG
    pf_stream = torch.cuda.Stream()
    filler_x = torch.randn(64, 4096, device=dev)
    for name, (N, K) in SHAPES.items():
        if a.only and name != a.only:
            continue
        KC, waves, sm = cfgs[name]
        x = torch.randn(32, K, device=dev).to(torch.bfloat16)
        wps = [ops.pack_weight((torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16), KC, sm) for _ in range(a.copies)]
        g1 = lambda i: ops.skinny_gemm(x, wps[i % a.copies], N, K, KC, waves, sm)
        r = dict(shape=name, MB=round(N * K * 2 / 1e6, 1))
        r["cold_us"] = round(graph_time(g1, a.launches), 2)
        for blocks in (64, 128, 256, 512, 1024):
            pf = lambda i, b=blocks: ops.weight_prefetch(wps[i % a.copies], b)
            t_pf = graph_time(pf, a.launches)
            both = lambda i, b=blocks: (ops.weight_prefetch(wps[i % a.copies], b), g1(i))
            t_both = graph_time(both, a.launches)
            r[f"pf{blocks}_us"] = round(t_pf, 2)
            r[f"pf{blocks}_TBps"] = round(N * K * 2 / 1e6 / t_pf, 3)
            r[f"hot_after_pf{blocks}_us"] = round(t_both - t_pf, 2)

        # overlap: the prefetch of copy i+1 runs on a side stream under a latency-bound filler (16 small dependent kernels ~ the
        # F2/K1/combine/F1r chain), then G1 streams copy i+1
        def filler():
            y = filler_x
            for _ in range(6):
                y = y * 1.0001
            return y

        def ov(i, blocks=128, prefetch=True):
            cur = torch.cuda.current_stream()
            if prefetch:
                pf_stream.wait_stream(cur)
                with torch.cuda.stream(pf_stream):
                    ops.weight_prefetch(wps[(i + 1) % a.copies], blocks)
            filler()
            g1(i + 1)
            if prefetch:
                cur.wait_stream(pf_stream)

        r["filler_plus_g1_us"] = round(graph_time(lambda i: ov(i, prefetch=False), a.launches), 2)
        for blocks in (64, 128, 256):
            r[f"filler_pf{blocks}_g1_us"] = round(graph_time(lambda i, b=blocks: ov(i, b, True), a.launches), 2)
        print(json.dumps(r), flush=True)
        del wps
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
