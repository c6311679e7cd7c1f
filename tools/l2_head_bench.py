#!/usr/bin/env python3
"""Round 5: what does a G1z / G1sz launch gain when the head of its weight stream is already in the XCDs' L2 (csrc/sjd_l2_prefetch.h)?
Per Lumina-7B projection shape, in hipGraphs over distinct weight sets (every launch streams from HBM):
    tiny_g1   n x [a pull of one pair by 8 workgroups -> G1z]     (the same number of kernel boundaries as the next line)
    pull_g1   n x [pull of `head_pairs` pairs per unit by `blocks` workgroups -> G1z]
    pull      n x [the pull alone]          tiny   n x [the tiny pull alone]
gain of the consumer = (pull - tiny) - (pull_g1 - tiny_g1): what the warm head takes off the G1z launch.
Also checks the dispatch rule the pull relies on: workgroup L of a launch runs on XCD L mod 8."""
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
    ap.add_argument("--rows", type=int, default=32)
    ap.add_argument("--only", default="")
    ap.add_argument("--pairs", default="4,8,12,16")
    ap.add_argument("--blocks", default="256,512")
    ap.add_argument("--between", type=int, default=0, help="small dependent launches between the pull and the consumer (F2 -> K1 -> combine -> o)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load_exp()
    import sjd_amd.backbones as BB
    ok = True
    for gx, gy in ((256, 1), (64, 4), (172, 1), (22, 8), (16, 15), (1024, 1), (3, 5)):
        m = ops.xcc_map(gx, gy).cpu().flatten()
        exp = torch.arange(gx * gy, dtype=torch.int32) % 8
        same = bool((m == exp).all())
        ok &= same
        print(json.dumps(dict(xcc_round_robin=[gx, gy], holds=same, first16=m[:16].tolist())), flush=True)
    shapes = dict(qkv=(12288, 4096), o=(4096, 4096), gate_up=(22016, 4096), down=(4096, 11008))
    small = torch.zeros(256 * 256, device=dev)
    for name, (N, K) in shapes.items():
        if a.only and name not in a.only.split(","):
            continue
        KC, waves, sm = BB.ChameleonBackbone.G1_CFG_Z[name]
        x = torch.randn(a.rows, K, device=dev).to(torch.bfloat16)
        wzs = [ops.pack_weight_z((torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16), KC, sm) for _ in range(a.copies)]
        gateup = name == "gate_up"

        def g1(i):
            if gateup:
                return ops.gateup_silu(x, wzs[i % a.copies], N // 2, K, sm)
            return ops.skinny_gemm(x, wzs[i % a.copies], N, K, KC, waves, sm)

        tiny = [ops.l2_head(z, a.rows, waves, 1, gateup=gateup) for z in wzs]

        def between():
            for _ in range(a.between):
                small.add_(1.0)

        t_g1 = timed_graph(lambda i: g1(i), a.launches, lib)[0] * 1e3
        t_tiny = timed_graph(lambda i: ops.weight_prefetch_head(tiny[i % a.copies], 8), a.launches, lib)[0] * 1e3
        t_tiny_g1 = timed_graph(lambda i: (ops.weight_prefetch_head(tiny[i % a.copies], 8), between(), g1(i)), a.launches, lib)[0] * 1e3
        base = dict(shape=name, N=N, K=K, cfg=[KC, waves, int(sm)], stored_MB=round(wzs[0].nbytes() / 1e6, 1), g1_us=round(t_g1, 2), tiny_us=round(t_tiny, 2),
                    tiny_g1_us=round(t_tiny_g1, 2), between=a.between)
        print(json.dumps(base), flush=True)
        for F in [int(v) for v in a.pairs.split(",")]:
            heads = [ops.l2_head(z, a.rows, waves, F, gateup=gateup) for z in wzs]
            mb = ops.l2_head_bytes(heads[0]) / 1e6
            for P in [int(v) for v in a.blocks.split(",")]:
                t_pull = timed_graph(lambda i: ops.weight_prefetch_head(heads[i % a.copies], P), a.launches, lib)[0] * 1e3
                t_pull_g1 = timed_graph(lambda i: (ops.weight_prefetch_head(heads[i % a.copies], P), between(), g1(i)), a.launches, lib)[0] * 1e3
                # the same pull on the WRONG XCDs (pairs of another weight copy's... no: the same bytes, pulled by workgroups shifted by one XCD)
                gain = (t_pull - t_tiny) - (t_pull_g1 - t_tiny_g1)
                print(json.dumps(dict(shape=name, head_pairs=F, head_MB=round(mb, 2), blocks=P, pull_us=round(t_pull, 2), pull_g1_us=round(t_pull_g1, 2),
                                      g1_after_pull_us=round(t_pull_g1 - t_pull, 2), consumer_gain_us=round(gain, 2))), flush=True)
        del wzs
        torch.cuda.empty_cache()
    print(json.dumps(dict(xcc_round_robin_holds=ok)))


if __name__ == "__main__":
    main()
