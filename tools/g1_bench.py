#!/usr/bin/env python3
"""Micro-driver of kernel G1 (weight-streaming projection, M=32) at the Lumina-mGPT-7B layer shapes, cycling over enough
weight copies that every launch streams from HBM (not the 256 MB Infinity Cache); also times hipBLASLt (F.linear) on the
same shapes.  Prints one JSON line per shape."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import sjd_amd._lib as L  # noqa: E402
import sjd_amd.ops as ops  # noqa: E402

SHAPES = dict(qkv=(12288, 4096, 1024), o=(4096, 4096, 512), gate_up=(22016, 4096, 2048), down=(4096, 11008, 1024))


def timed_batched(fn, n, lib):
    """n back-to-back launches between ONE event pair: the command processor pipelines the dispatches, as inside a hipGraph."""
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    hip = ctypes.CDLL("libamdhip64.so")
    e0, e1 = ctypes.c_void_p(lib.sjd_event_create()), ctypes.c_void_p(lib.sjd_event_create())
    res = []
    for rep in range(3):
        hip.hipEventRecord(e0, stream)
        for i in range(n):
            fn(i)
        hip.hipEventRecord(e1, stream)
        torch.cuda.synchronize()
        res.append(lib.sjd_event_elapsed_ms(e0, e1) / n)
    return min(res), sorted(res)[1]


def timed_graph(fn, n, lib):
    """n launches captured in one hipGraph and replayed: no host launch cost, the conditions of the engine's forward graph."""
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
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    return min(res), sorted(res)[1]


def timed(fn, n, lib):
    evs = [(ctypes.c_void_p(lib.sjd_event_create()), ctypes.c_void_p(lib.sjd_event_create())) for _ in range(n)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    hip = ctypes.CDLL("libamdhip64.so")
    for i, (e0, e1) in enumerate(evs):
        hip.hipEventRecord(e0, stream)
        fn(i)
        hip.hipEventRecord(e1, stream)
    torch.cuda.synchronize()
    ms = sorted(lib.sjd_event_elapsed_ms(e0, e1) for e0, e1 in evs)
    return sum(ms) / len(ms), ms[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=64)
    ap.add_argument("--copies", type=int, default=12)
    ap.add_argument("--only", default="")
    ap.add_argument("--kc", type=int, default=0)
    ap.add_argument("--waves", type=int, default=4)
    ap.add_argument("--no-blas", action="store_true")
    ap.add_argument("--step-major", type=int, default=0)
    ap.add_argument("--batched", action="store_true", help="back-to-back eager launches, one event pair (host launch rate bound for short kernels)")
    ap.add_argument("--per-launch", action="store_true", help="one event pair per launch (includes ~4 us dispatch latency)")
    ap.add_argument("--product", action="store_true", help="per-shape (KC, waves, layout) of backbones.G1_CFG")
    ap.add_argument("--sweep", action="store_true", help="grid over KC x waves x layout per shape (one JSON line per point, best first at the end)")
    ap.add_argument("--fine", action="store_true", help="with --sweep: KC in steps of 64 and waves 6..12, one layout per shape")
    ap.add_argument("--tiled64", action="store_true", help="with --sweep --rows 64: include KC > 1280 (served by the sub-tiled kernel, <= 8 waves)")
    ap.add_argument("--rows", type=int, default=32, help="window rows (64: draft window 32 or two prompts; the staged chunk must then be <= 1280 columns)")
    ap.add_argument("--emu3", action="store_true", help="Emu3-Gen 8B projection shapes (GQA 32/8, intermediate 14336) instead of Lumina-7B")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if os.environ.get("SJD_SO"):
        L.SO_PATH = os.environ["SJD_SO"]
    lib = L.load()
    if a.emu3:
        SHAPES.update(qkv=(6144, 4096, 512), o=(4096, 4096, 512), gate_up=(28672, 4096, 1024), down=(4096, 14336, 1024))
    if a.sweep:
        for name, (N, K, _) in SHAPES.items():
            if a.only and name != a.only:
                continue
            x = torch.randn(a.rows, K, device=dev).to(torch.bfloat16)
            ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(a.copies)]
            rows = []
            for sm in ((1,) if name in ("qkv", "gate_up") else (0,)) if a.fine else (1, 0):
                for KC in (range(512, 2049, 64) if a.fine else (256, 512, 688, 896, 1024, 1280, 1376, 1536, 2048)):
                    if KC > K or (name == "down" and KC == 2048) or (32 < a.rows <= 64 and KC > 1280 and not a.tiled64) or K % 16:
                        continue
                    wps = [ops.pack_weight(w, KC, bool(sm)) for w in ws]
                    nc = (K + KC - 1) // KC
                    out = torch.empty(nc, ((a.rows + 31) // 32) * 32, N, dtype=torch.float32, device=dev)
                    for waves in ((6, 7, 8, 9, 10, 12) if a.fine else (4, 6, 8, 11, 12, 16)):
                        if (a.rows > 64 or (a.tiled64 and KC > 1280)) and waves > 8:
                            continue
                        n_wg = ((N // 32 + waves - 1) // waves) * nc

                        def g1(i, wps=wps, KC=KC, waves=waves, sm=sm, out=out):
                            L.check(lib.sjd_skinny_gemm(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wps[i % a.copies].data_ptr()),
                                                        ctypes.c_void_p(out.data_ptr()), a.rows, N, K, KC, waves, sm, 0,
                                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "g1")
                        avg, _ = timed_graph(g1, a.launches, lib)
                        r = dict(shape=name, KC=KC, waves=waves, step_major=sm, workgroups=n_wg, us=round(avg * 1e3, 2),
                                 TBps=round(N * K * 2 / 1e12 / (avg / 1e3), 3))
                        rows.append(r)
                        print(json.dumps(r), flush=True)
                    del wps, out
                    torch.cuda.empty_cache()
            best = sorted(rows, key=lambda r: r["us"])[:5]
            print(json.dumps(dict(shape=name, best=best)), flush=True)
        return
    for name, (N, K, KC) in SHAPES.items():
        if a.only and name != a.only:
            continue
        KC = a.kc or KC
        if a.product:
            import sjd_amd.backbones as BB
            cfg = BB.ChameleonBackbone.G1_CFG if a.rows <= 32 else BB.ChameleonBackbone.G1_CFG_64ROW if a.rows <= 64 else BB.ChameleonBackbone.G1_CFG_128ROW
            KC, a.waves, sm = cfg[name]
            a.step_major = int(sm)
        x = torch.randn(a.rows, K, device=dev).to(torch.bfloat16)
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(a.copies)]
        wps = [ops.pack_weight(w, KC, bool(a.step_major)) for w in ws]
        nc = (K + KC - 1) // KC
        out = torch.empty(nc, ((a.rows + 31) // 32) * 32, N, dtype=torch.float32, device=dev)

        def g1(i):
            L.check(lib.sjd_skinny_gemm(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wps[i % a.copies].data_ptr()),
                                        ctypes.c_void_p(out.data_ptr()), a.rows, N, K, KC, a.waves, a.step_major, 0,
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "g1")

        def blas(i):
            F.linear(x, ws[i % a.copies])

        for f in (g1, blas):
            for i in range(a.copies):
                f(i)
        torch.cuda.synchronize()
        bytes_w = N * K * 2
        r = dict(shape=name, N=N, K=K, KC=KC, weight_MB=round(bytes_w / 1e6, 1))
        r["waves"], r["step_major"] = a.waves, a.step_major
        for tag, f in ((("g1", g1),) if a.no_blas else (("g1", g1), ("hipblaslt", blas))):
            avg, med = (timed if a.per_launch else timed_batched if a.batched else timed_graph)(f, a.launches, lib)
            r[tag + "_us"] = round(avg * 1e3, 2)
            r[tag + "_TBps"] = round(bytes_w / 1e12 / (avg / 1e3), 3)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
