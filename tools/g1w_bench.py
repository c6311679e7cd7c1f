#!/usr/bin/env python3
"""G1w (csrc/sjd_gemm_wide.h: the 65..256-row weight-streaming projection, round 6) against g1_skinny_gemm_tiled8 and hipBLASLt at the
Lumina-7B layer shapes: hipGraph replays over several weight copies (every launch streams from HBM).  --check compares the planes bit for
bit with the 32-row kernel first.  One JSON line per (shape, configuration)."""
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

SHAPES = dict(qkv=(12288, 4096), o=(4096, 4096), gate_up=(22016, 4096), down=(4096, 11008))
# (KC, column tiles per workgroup, step-major) candidates per shape for the wide kernel
CAND = dict(qkv=[(2048, 3, 1), (2048, 4, 1), (2048, 8, 1), (1024, 8, 1), (832, 8, 1), (2048, 6, 1), (1376, 4, 1)],
            o=[(1024, 2, 1), (896, 4, 1), (512, 4, 1), (1024, 4, 1), (2048, 2, 1), (1024, 3, 1)],
            gate_up=[(2048, 4, 1), (2048, 8, 1), (1376, 8, 1), (2048, 6, 1), (1376, 4, 1), (1024, 8, 1)],
            down=[(2752, 2, 1), (1408, 4, 1), (2752, 4, 1), (1376, 4, 1), (2752, 3, 1), (1408, 8, 1), (2752, 8, 1)])
OLD = dict(qkv=(2048, 4, 1), o=(896, 4, 1), gate_up=(2048, 4, 1), down=(1376, 4, 1))          # backbones.G1_CFG_256ROW (round 5)
OLD128 = dict(qkv=(2048, 4, 1), o=(896, 4, 1), gate_up=(2048, 8, 1), down=(1376, 4, 1))       # backbones.G1_CFG_128ROW
VP = ctypes.c_void_p


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
    ap.add_argument("--rows", type=int, default=256)
    ap.add_argument("--launches", type=int, default=24)
    ap.add_argument("--copies", type=int, default=6)
    ap.add_argument("--only", default="")
    ap.add_argument("--variants", default="0", help="comma list of variants of sjd_skinny_gemm_wide (csrc/sjd_gemm.hip): 0 = the product's (stage 4 k-steps, 3 slots), 1 (4, 4), 10 eight waves, 20 two workgroups per CU (128 rows)")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--no-old", action="store_true")
    ap.add_argument("--no-blas", action="store_true")
    ap.add_argument("--pad", type=int, default=0, help="row stride of x = K + pad elements (L2 channel spread experiment)")
    ap.add_argument("--trace", action="store_true", help="SJD_HIP_EXP_LIB is a -DSJD_TRACE build: per-workgroup wall / shader-clock stamps of the last launch")
    ap.add_argument("--cand", default="", help="KC:tiles:step_major[,...] instead of the built-in candidates")
    ap.add_argument("--emu3", action="store_true", help="Emu3-8B projection shapes (q|k|v 6144, o 4096, gate|up 28672 x 4096, down 4096 x 14336); use with --rows 64")
    a = ap.parse_args()
    if a.emu3:
        SHAPES.update(qkv=(6144, 4096), o=(4096, 4096), gate_up=(28672, 4096), down=(4096, 14336))
        CAND.update(qkv=[(512, 8, 0), (512, 4, 1), (1024, 4, 1), (1024, 8, 1), (2048, 4, 1), (2048, 3, 1), (832, 6, 1)],
                    o=[(512, 8, 0), (512, 4, 1), (1024, 2, 1), (1024, 4, 1), (512, 2, 1)],
                    gate_up=[(2048, 8, 1), (2048, 6, 1), (2048, 4, 1), (1024, 8, 1), (4096, 4, 1)],
                    down=[(896, 8, 0), (896, 8, 1), (1792, 4, 1), (1024, 4, 1), (2048, 4, 1), (1792, 8, 1)])
        OLD128.update(qkv=(512, 8, 0), o=(512, 8, 0), gate_up=(2048, 8, 1), down=(896, 8, 0))      # backbones.G1_CFG_EMU3 (64-row windows)
    dev = torch.device("cuda:0")
    lib = L.load_exp()          # (the tuning entry sjd_skinny_gemm_wide lives in libsjd_hip_exp.so; SJD_HIP_EXP_LIB: a probe build of it)
    M = a.rows
    stream = lambda: VP(torch.cuda.current_stream().cuda_stream)
    for name, (N, K) in SHAPES.items():
        if a.only and name not in a.only.split(","):
            continue
        g = torch.Generator(device="cpu").manual_seed(N + K)
        xfull = torch.zeros(M, K + a.pad, dtype=torch.bfloat16, device=dev)
        xfull[:, :K] = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        x = xfull[:, :K] if a.pad else xfull
        xc = x.contiguous()
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(a.copies)]
        bytes_w = N * K * 2
        cands = [tuple(int(v) for v in c.split(":")) for c in a.cand.split(",")] if a.cand else CAND[name]
        if not a.no_blas:
            us = timed_graph(lambda i: F.linear(xc, ws[i % a.copies]), a.launches)
            print(json.dumps(dict(shape=name, rows=M, kernel="hipblaslt", us=round(us, 2), TBps=round(bytes_w / us / 1e6, 3))), flush=True)
        if not a.no_old:
            KC, waves, sm = (OLD if M > 128 else OLD128)[name]
            wps = [ops.pack_weight(w, KC, bool(sm)) for w in ws]
            nc = (K + KC - 1) // KC
            out = torch.empty(nc, ((M + 31) // 32) * 32, N, dtype=torch.float32, device=dev)
            os.environ["SJD_G1_WIDE"] = "0"

            def old(i):
                L.check(lib.sjd_skinny_gemm(VP(xc.data_ptr()), VP(wps[i % a.copies].data_ptr()), VP(out.data_ptr()), M, N, K, KC, waves, sm, 0, stream()), "g1")
            us = timed_graph(old, a.launches)
            print(json.dumps(dict(shape=name, rows=M, kernel="tiled8", KC=KC, tiles=waves, planes=nc, us=round(us, 2), TBps=round(bytes_w / us / 1e6, 3))), flush=True)
            del wps, out
        rows = []
        for (KC, tiles, sm) in cands:
            wps = [ops.pack_weight(w, KC, bool(sm)) for w in ws]
            nc = (K + KC - 1) // KC
            out = torch.empty(nc, ((M + 31) // 32) * 32, N, dtype=torch.float32, device=dev)
            n_wg = ((N // 32 + tiles - 1) // tiles) * nc
            for var in [int(v) for v in a.variants.split(",")]:
                def wide(i, var=var):
                    L.check(lib.sjd_skinny_gemm_wide(VP(x.data_ptr()), VP(wps[i % a.copies].data_ptr()), VP(out.data_ptr()), M, N, K, KC, tiles, sm, var, K + a.pad, stream()), "g1w")
                r = dict(shape=name, rows=M, kernel="wide", KC=KC, tiles=tiles, step_major=sm, variant=var, planes=nc, workgroups=n_wg)
                if a.check and KC <= 2560:      # (the 32-row reference stages a whole chunk in LDS)
                    out.fill_(float("nan"))
                    wide(0)
                    torch.cuda.synchronize()
                    ref = xc.float() @ ws[0].float().t()
                    got = out.sum(0)[:M]
                    r["max_err"] = float((got - ref).abs().max())
                    same = True
                    for r0 in range(0, M, 32):
                        p32 = ops.skinny_gemm(xc[r0:r0 + 32].contiguous(), wps[0], N, K, KC, waves=4, step_major=bool(sm))
                        nr = min(32, M - r0)
                        same = same and bool(torch.equal(p32.data[:, :nr], out[:, r0:r0 + nr]))
                    r["bit_identical_to_32row_kernel"] = same
                    if M < out.shape[1]:
                        r["pad_rows_zero"] = bool(out[:, M:].abs().max() == 0)
                us = timed_graph(wide, a.launches)
                r.update(us=round(us, 2), TBps=round(bytes_w / us / 1e6, 3))
                if a.trace:
                    import numpy as np
                    lib.sjd_debug_trace_g1.argtypes = [VP, ctypes.c_int]
                    buf = np.zeros((n_wg, 8), dtype=np.uint64)
                    assert lib.sjd_debug_trace_g1(buf.ctypes.data, n_wg) == 0
                    t = buf.astype(np.int64)
                    t0 = t[:, 0].min()
                    tick = 0.01          # 100 MHz wall clock -> us
                    loop_us = (t[:, 3] - t[:, 0]) * tick
                    mhz = (t[:, 5] - t[:, 4]) / np.maximum(loop_us, 1e-3)
                    r["trace"] = dict(start_skew_us=round(float((t[:, 0] - t0).max() * tick), 2), prologue_us=round(float(((t[:, 1] - t[:, 0]) * tick).mean()), 2),
                                      loop_us_mean=round(float(((t[:, 3] - t[:, 1]) * tick).mean()), 2), loop_us_max=round(float(((t[:, 3] - t[:, 1]) * tick).max()), 2),
                                      stores_issued_us=round(float(((t[:, 2] - t[:, 3]) * tick).mean()), 2), stores_acked_us=round(float(((t[:, 6] - t[:, 2]) * tick).mean()), 2),
                                      end_mean_us=round(float(((t[:, 6] - t0) * tick).mean()), 2), end_max_us=round(float(((t[:, 6] - t0) * tick).max()), 2),
                                      shader_mhz_mean=round(float(mhz.mean()), 0), shader_mhz_min=round(float(mhz.min()), 0))
                rows.append(r)
                print(json.dumps(r), flush=True)
            del wps, out
            torch.cuda.empty_cache()
        if rows:
            print(json.dumps(dict(shape=name, best=sorted(rows, key=lambda r: r["us"])[:3])), flush=True)


if __name__ == "__main__":
    main()
