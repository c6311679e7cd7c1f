#!/usr/bin/env python3
"""Micro-driver of kernel K1 (draft-window attention) at the Lumina-mGPT-7B shape, cycling over the 32 per-layer KV
caches exactly like one SJD iteration does (so the working set, ~40 MB/layer x 32, streams from HBM instead of sitting
in the 256 MB Infinity Cache).  Used under rocprofv3 (--kernel-trace --stats, and --pmc FETCH_SIZE / WRITE_SIZE in
separate passes) to produce profiles/*k1*.  Prints one JSON line with the HIP-event timing."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sjd_amd._lib as L  # noqa: E402
import sjd_amd.ops as ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kv-len", type=int, default=1216)
    ap.add_argument("--launches", type=int, default=320)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=32)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--window", type=int, default=16)
    ap.add_argument("--n-split", type=int, default=8)
    ap.add_argument("--prompt", type=int, default=64)
    ap.add_argument("--fp8", action="store_true", help="fp8 (e4m3) KV cache + the fp8-MFMA K1 variant (BASELINE config 5)")
    ap.add_argument("--colsplit", action="store_true", help="the column-split form (no key splits, no combine; multi-head 16-row windows)")
    ap.add_argument("--graph", action="store_true", help="time k1_partial + k1_combine per layer inside one hipGraph replay")
    ap.add_argument("--block", choices=["fused", "unfused"], default=None,
                    help="time the whole attention block of a layer on q|k|v split-K partials inside a hipGraph: fused = kernel K1F (one "
                         "launch), unfused = F2 + k1_partial + k1_combine (three launches)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load()
    B, n, H, Hkv, D = 2, a.window, a.heads, a.kv_heads, a.head_dim
    s_max = ((a.kv_len + n + 64 + 31) // 32) * 32
    kc = torch.randn(a.layers, B, Hkv, s_max, D, device=dev).to(torch.bfloat16)
    vc = torch.randn(a.layers, B, Hkv, s_max, D, device=dev).to(torch.bfloat16)
    q = torch.randn(B, n, H, D, device=dev).to(torch.bfloat16)
    out = torch.empty_like(q)
    ks = torch.tensor([0, a.prompt - 1], dtype=torch.int32, device=dev)
    ws = ops.attention_workspace(B, H, n, D, a.n_split, dev)
    if a.fp8:
        kc, vc = kc.to(ops.FP8), vc.to(ops.FP8)
    if a.block:
        n_chunks = 4
        parts = [ops.Partials(torch.randn(n_chunks, 32, (H + 2 * Hkv) * D, device=dev), n_chunks, (H + 2 * Hkv) * D) for _ in range(4)]
        mk = lambda m_, s_: (m_ + s_ * torch.randn(1, D, device=dev)).to(torch.bfloat16)
        qn = (mk(1, .1), mk(0, .1), mk(1, .1), mk(0, .1))
        inv = (1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))).to(dev)
        pos = (a.kv_len + torch.arange(n, device=dev)[None] - ks[:, None].long()).reshape(-1).contiguous()
        rn = (torch.full((8, 32), 512.0, device=dev), 4096, 1e-5)
        attn = ops.HipWindowAttention(n_split=a.n_split)

        class C:
            pass
        cache = C()
        cache.k, cache.v = kc, vc

        def one(i):
            if a.block == "fused":
                ops.qkv_attention_fused(parts[i % 4], kc[i], vc[i], *qn, inv, pos, B, n, H, D, None, a.kv_len, ks, row_norm=rn, dtype=torch.bfloat16)
            else:
                q_ = ops.qknorm_rope_append(parts[i % 4], kc[i], vc[i], *qn, inv, pos, B, n, H, Hkv, D, None, a.kv_len, dtype=torch.bfloat16, row_norm=rn)
                attn.attend(i, q_, cache, a.kv_len, ks)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            one(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(a.layers):
                one(i)
        g.replay()
        torch.cuda.synchronize()
        reps = max(1, a.launches // a.layers)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        avg = e0.elapsed_time(e1) / (reps * a.layers)
        rows0, rows1 = a.kv_len + n, a.kv_len + n - (a.prompt - 1)
        alg = 2 * Hkv * (rows0 + rows1) * D * 2 + n_chunks * B * n * (H + 2 * Hkv) * D * 4 + 2 * B * n * Hkv * D * 2 + B * n * H * D * 2
        print(json.dumps(dict(kernel=f"attention block, {a.block} (hipGraph replay)", kv_len=a.kv_len, window=n, n_split=a.n_split,
                              launches=reps * a.layers, avg_us=round(avg * 1e3, 2), algorithmic_bytes=alg,
                              gbps=round(alg / 1e9 / (avg / 1e3), 1), frac_of_8TBps=round(alg / 1e9 / (avg / 1e3) / 8000, 4))))
        return
    if a.fp8 or a.graph:
        def one(i):
            if a.colsplit:
                ops.draft_window_attention_colsplit(q, kc[i], vc[i], out, ks, None, a.kv_len)
            elif a.fp8:
                ops.draft_window_attention_fp8(q, kc[i], vc[i], out, 1.0, 1.0, ks, None, a.kv_len, a.n_split, ws)
            else:
                ops.draft_window_attention(q, kc[i], vc[i], out, ks, None, a.kv_len, a.n_split, ws)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            one(0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(a.layers):
                one(i)
        g.replay()
        torch.cuda.synchronize()
        reps = max(1, a.launches // a.layers)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        avg = e0.elapsed_time(e1) / (reps * a.layers)
        esz = kc.element_size()
        rows0, rows1 = a.kv_len + n, a.kv_len + n - (a.prompt - 1)
        alg = 2 * Hkv * (rows0 + rows1) * D * esz + B * n * H * D * 2
        print(json.dumps(dict(kernel="K1 per layer (hipGraph replay): the form the launcher picks for the shape -- k1_partial / k1_partial_ring + k1_combine, or the column split", kv=("fp8" if a.fp8 else "bf16"), kv_len=a.kv_len, window=n,
                              launches=reps * a.layers, avg_us=round(avg * 1e3, 2), algorithmic_bytes=alg,
                              gbps=round(alg / 1e9 / (avg / 1e3), 1), frac_of_8TBps=round(alg / 1e9 / (avg / 1e3) / 8000, 4))))
        return
    for i in range(a.layers):
        ops.draft_window_attention(q, kc[i], vc[i], out, ks, None, a.kv_len, a.n_split, ws)
    torch.cuda.synchronize()
    evs = [(ctypes.c_void_p(lib.sjd_event_create()), ctypes.c_void_p(lib.sjd_event_create())) for _ in range(a.launches)]
    for i, (e0, e1) in enumerate(evs):
        ops.draft_window_attention(q, kc[i % a.layers], vc[i % a.layers], out, ks, None, a.kv_len, a.n_split, ws, e0, e1)
    torch.cuda.synchronize()
    ms = sorted(lib.sjd_event_elapsed_ms(e0, e1) for e0, e1 in evs)
    esz = 2
    rows0, rows1 = a.kv_len + n, a.kv_len + n - (a.prompt - 1)
    alg = 2 * Hkv * (rows0 + rows1) * D * esz + B * n * H * D * esz
    avg = sum(ms) / len(ms)
    print(json.dumps(dict(kernel="k1_partial", kv_len=a.kv_len, window=n, launches=len(ms), avg_us=round(avg * 1e3, 2),
                          median_us=round(ms[len(ms) // 2] * 1e3, 2), algorithmic_bytes=alg,
                          gbps=round(alg / 1e9 / (avg / 1e3), 1), frac_of_8TBps=round(alg / 1e9 / (avg / 1e3) / 8000, 4))))


if __name__ == "__main__":
    main()
