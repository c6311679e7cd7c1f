#!/usr/bin/env python3
"""What the latency-floor kernels of a layer (F1r, F2, F3, k1_combine) cost as dependent stages of a hipGraph chain, alone and behind
the G1 launch that produces their input, at the Lumina-7B shapes.  tools/stage_floor_probe.hip puts the floor of a dependent stage that
reads 8 MB written elsewhere and writes 1 MB at ~2.4 us; rocprofv3 shows these kernels at 4.6-5.4 us inside the forward.  One JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sjd_amd._lib as L  # noqa: E402
import sjd_amd.ops as ops  # noqa: E402
import sjd_amd.backbones as BB  # noqa: E402


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
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n * 1e3)
    return round(sorted(res)[2], 2)


def main():
    dev = torch.device("cuda:0")
    L.load()
    T, hid, inter, H, D = 32, 4096, 11008, 32, 128
    dt = torch.bfloat16
    N = 64
    cfg = BB.ChameleonBackbone.G1_CFG
    h = torch.randn(T, hid, device=dev).to(dt)
    mk_part = lambda nc, n: ops.Partials(torch.randn(nc, 32, n, device=dev) * 0.1, nc, n)
    p_down, p_o, p_gu, p_qkv = mk_part(11, hid), mk_part(8, hid), mk_part(2, 2 * inter), mk_part(4, 3 * H * D)
    sumsq = torch.full((8, 32), 512.0, device=dev)
    rn = (sumsq, hid, 1e-5)
    kc = torch.zeros(2, H, 256, D, device=dev, dtype=dt)
    vc = torch.zeros_like(kc)
    inv = (1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))).to(dev)
    pos = torch.arange(T, device=dev)
    mk = lambda m_, s_: (m_ + s_ * torch.randn(1, D, device=dev)).to(dt)
    qn = (mk(1, .1), mk(0, .1), mk(1, .1), mk(0, .1))
    r = {}
    r["f1r_11chunks"] = graph_time(lambda i: ops.residual_sumsq(h, p_down), N)
    r["f1r_8chunks"] = graph_time(lambda i: ops.residual_sumsq(h, p_o), N)
    r["f1r_nopart"] = graph_time(lambda i: ops.residual_sumsq(h, None), N)
    r["f3"] = graph_time(lambda i: ops.silu_mul(p_gu, rows=T, dtype=dt, row_norm=rn), N)
    r["f2"] = graph_time(lambda i: ops.qknorm_rope_append(p_qkv, kc, vc, *qn, inv, pos, 2, 16, H, H, D, None, 100, dtype=dt, row_norm=rn), N)
    # behind their producer: G1 + consumer minus G1 alone (weights cycle over 6 copies so that G1 streams from HBM)
    def packed(Nn, K, name):
        return [ops.pack_weight((torch.randn(Nn, K, device=dev) / K ** 0.5).to(dt), cfg[name][0], cfg[name][2]) for _ in range(6)]
    x_h = torch.randn(T, hid, device=dev).to(dt)
    x_i = torch.randn(T, inter, device=dev).to(dt)
    for name, Nn, K, x, cons in (("down", hid, inter, x_i, lambda p: ops.residual_sumsq(h, p)),
                                 ("o", hid, hid, x_h, lambda p: ops.residual_sumsq(h, p)),
                                 ("gate_up", 2 * inter, hid, x_h, lambda p: ops.silu_mul(p, rows=T, dtype=dt, row_norm=rn)),
                                 ("qkv", 3 * H * D, hid, x_h, lambda p: ops.qknorm_rope_append(p, kc, vc, *qn, inv, pos, 2, 16, H, H, D, None, 100, dtype=dt, row_norm=rn))):
        wps = packed(Nn, K, name)
        g1 = lambda i: ops.skinny_gemm(x, wps[i % 6], Nn, K, cfg[name][0], cfg[name][1], cfg[name][2])
        a = graph_time(lambda i: g1(i), 36)
        b = graph_time(lambda i: cons(g1(i)), 36)
        r[f"g1_{name}"] = a
        r[f"g1_{name}_plus_consumer"] = b
        r[f"consumer_after_{name}"] = round(b - a, 2)
        del wps
        torch.cuda.empty_cache()
    print(json.dumps(r))


if __name__ == "__main__":
    main()
