#!/usr/bin/env python3
"""Round 5: net effect of HOSTING the L2 head pull in F1r.  One iteration = down G1z -> F1r (residual + row statistics) -> q|k|v G1z, the
F1r launch with / without the pull of the q|k|v head; and o G1z -> F1r -> gate|up G1sz likewise.  hipGraph over distinct weight sets."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import sjd_amd._lib as L  # noqa: E402
import sjd_amd.ops as ops  # noqa: E402
from g1_bench import timed_graph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=32)
    ap.add_argument("--copies", type=int, default=8)
    ap.add_argument("--pairs", default="4,8,12")
    ap.add_argument("--blocks", default="512,1024,2048")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load_exp()
    import sjd_amd.backbones as BB
    cfg = BB.ChameleonBackbone.G1_CFG_Z
    rows, hid, inter = 32, 4096, 11008
    rnd = lambda n, k: (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    for first, second in (("down", "qkv"), ("o", "gate_up")):
        shp = dict(qkv=(12288, hid), o=(hid, hid), gate_up=(2 * inter, hid), down=(hid, inter))
        (N1, K1), (N2, K2) = shp[first], shp[second]
        w1 = [ops.pack_weight_z(rnd(N1, K1), cfg[first][0], cfg[first][2]) for _ in range(a.copies)]
        w2 = [ops.pack_weight_z(rnd(N2, K2), cfg[second][0], cfg[second][2]) for _ in range(a.copies)]
        x1 = torch.randn(rows, K1, device=dev).to(torch.bfloat16)
        h = torch.randn(rows, hid, device=dev).to(torch.bfloat16)
        gateup = second == "gate_up"

        def it(i, head, blocks):
            part = ops.skinny_gemm(x1, w1[i % a.copies], N1, K1, cfg[first][0], cfg[first][1], cfg[first][2])
            ss = ops.residual_sumsq(h, part, pull=head, pull_blocks=blocks)
            if gateup:
                return ops.gateup_silu(h, w2[i % a.copies], inter, hid, cfg[second][2], row_norm=(ss, hid, 1e-5))
            return ops.skinny_gemm(h, w2[i % a.copies], N2, K2, cfg[second][0], cfg[second][1], cfg[second][2])

        t0 = timed_graph(lambda i: it(i, None, 0), a.launches, lib)[0] * 1e3
        t0b = timed_graph(lambda i: it(i, None, 0), a.launches, lib)[0] * 1e3
        print(json.dumps(dict(chain=f"{first} -> F1r -> {second}", no_pull_us=[round(t0, 2), round(t0b, 2)])), flush=True)
        for F in [int(v) for v in a.pairs.split(",")]:
            heads = [ops.l2_head(z, rows, cfg[second][1], F, gateup=gateup) for z in w2]
            for P in [int(v) for v in a.blocks.split(",")]:
                t = timed_graph(lambda i: it(i, heads[i % a.copies], P), a.launches, lib)[0] * 1e3
                print(json.dumps(dict(chain=f"{first} -> F1r+pull -> {second}", head_pairs=F, head_MB=round(ops.l2_head_bytes(heads[0]) / 1e6, 1), pull_blocks=P,
                                      us=round(t, 2), gain_us=round(min(t0, t0b) - t, 2))), flush=True)
        del w1, w2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
