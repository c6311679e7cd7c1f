#!/usr/bin/env python3
"""G1sz (gate|up + SiLU * up) -> G1z (down) as two launches against sjd_mlp_pair_z (one launch, down's weight stream started ahead of the
dependency edge): us per MLP inside a hipGraph over `--layers` distinct weight sets (every launch streams from HBM).  Round-4 go / no-go."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import sjd_amd.ops as ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--inter", type=int, default=11008)
    ap.add_argument("--kc-down", type=int, default=768)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    hid, I = 4096, a.inter
    g = torch.Generator().manual_seed(0)
    x = torch.randn(32, hid, generator=g).to(torch.bfloat16).to(dev)
    rn = (ops.residual_sumsq(x.clone(), None), hid, 1e-5)
    ws = []
    for li in range(a.layers):
        wgu = (torch.randn(2 * I, hid, generator=g) * 0.02).to(torch.bfloat16).to(dev)
        wdn = (torch.randn(hid, I, generator=g) * 0.02).to(torch.bfloat16).to(dev)
        ws.append((ops.pack_weight_z(wgu, hid // 2, True), ops.pack_weight_z(wdn, a.kc_down, False)))
        del wgu, wdn
    assert ops.mlp_pair_ok(32, I, hid, ws[0][0], ws[0][1], a.kc_down, 8, dev)

    def two(li):
        y = ops.gateup_silu(x, ws[li][0], I, hid, True, row_norm=rn)
        return ops.skinny_gemm(y, ws[li][1], hid, I, a.kc_down, 8, False)

    def one(li):
        return ops.mlp_pair(x, ws[li][0], ws[li][1], I, hid, a.kc_down, row_norm=rn)[1]

    res = {}
    for name, fn in (("two_launches", two), ("pair", one), ("two_launches_again", two), ("pair_again", one)):
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            fn(0)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            keep = [fn(li) for li in range(a.layers)]
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) * 1e3 / (a.reps * a.layers), 2)
        del gr, keep
    res["timeouts"] = ops.mlp_pair_timeouts()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
