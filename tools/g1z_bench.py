#!/usr/bin/env python3
"""G1 / G1s against G1z / G1sz (the 12-bit lossless weight stream) at the product launch shapes of the Lumina-7B layer, launches
replayed from a hipGraph over enough weight copies that every launch streams from HBM.  One JSON line per shape: microseconds per
launch, bytes streamed, TB/s on the bytes actually moved."""
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
    ap.add_argument("--launches", type=int, default=48)
    ap.add_argument("--copies", type=int, default=8)
    ap.add_argument("--rows", type=int, default=32)
    ap.add_argument("--only", default="")
    ap.add_argument("--check", action="store_true", help="compare the results bit for bit before timing")
    ap.add_argument("--emu3", action="store_true", help="Emu3-8B projection shapes (q|k|v 6144 x 4096, o 4096 x 4096, down 4096 x 14336); use with --rows 64")
    ap.add_argument("--sweep", action="store_true", help="G1z only: grid over KC x waves x layout per shape (q|k|v, o, down), best five at the end")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load()
    import sjd_amd.backbones as BB
    cfg = BB.ChameleonBackbone.G1_CFG
    shapes = dict(qkv=(12288, 4096), o=(4096, 4096), gate_up=(22016, 4096), down=(4096, 11008))
    if a.emu3:
        shapes = dict(qkv=(6144, 4096), o=(4096, 4096), gate_up=(28672, 4096), down=(4096, 14336))
    if a.sweep:
        for name, (N, K) in shapes.items():
            if (a.only and name != a.only) or name == "gate_up":
                continue
            x = torch.randn(a.rows, K, device=dev).to(torch.bfloat16)
            ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(a.copies)]
            rows = []
            for sm in (True, False):
                for KC in (512, 640, 768, 896, 1024, 1280, 1536, 2048):
                    if KC > K:
                        continue
                    wzs = [ops.pack_weight_z(w, KC, sm) for w in ws]
                    for waves in (4, 6, 8, 12, 16):
                        if a.rows > 32 and (waves > 8 and (a.rows > 64 or KC > 1280)):
                            continue
                        avg, _ = timed_graph(lambda i: ops.skinny_gemm(x, wzs[i % a.copies], N, K, KC, waves, sm), a.launches, lib)
                        r = dict(shape=name, KC=KC, waves=waves, step_major=int(sm), workgroups=-(-(N // 32) // waves) * -(-K // KC), us=round(avg * 1e3, 2))
                        rows.append(r)
                        print(json.dumps(r), flush=True)
                    del wzs
                    torch.cuda.empty_cache()
            print(json.dumps(dict(shape=name, best=sorted(rows, key=lambda r: r["us"])[:5])), flush=True)
        return
    for name, (N, K) in shapes.items():
        if a.only and name != a.only:
            continue
        KC, waves, sm = cfg[name]
        KCz, wavesz, smz = BB.ChameleonBackbone.G1_CFG_Z[name]       # the launch shape the product uses for the 12-bit stream
        x = torch.randn(a.rows, K, device=dev).to(torch.bfloat16)
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(a.copies)]
        wps = [ops.pack_weight(w, KC, sm) for w in ws]
        wzs = [ops.pack_weight_z(w, KCz, smz) for w in ws]
        assert all(z is not None for z in wzs)
        exc = sum(z.n_exceptions for z in wzs) / a.copies
        del ws
        fused = name == "gate_up" and a.rows <= 32

        def run(pk, i):
            z = pk is wzs
            if fused:
                return ops.gateup_silu(x, pk[i % a.copies], N // 2, K, smz if z else sm)
            return ops.skinny_gemm(x, pk[i % a.copies], N, K, KCz if z else KC, wavesz if z else waves, smz if z else sm).data

        if a.check:
            for i in range(a.copies):
                r0, r1 = run(wps, i), run(wzs, i)
                torch.cuda.synchronize()
                if r0.dtype == torch.float32 and (KC, sm) != (KCz, smz):      # the two launch shapes split K differently: compare the summed planes
                    assert torch.allclose(r0.sum(0), r1.sum(0), rtol=1e-4, atol=1e-4), name
                else:
                    assert torch.equal(r0.view(torch.int32 if r0.dtype == torch.float32 else torch.int16),
                                       r1.view(torch.int32 if r1.dtype == torch.float32 else torch.int16)), name
        r = dict(shape=name, kernel="G1s" if fused else "G1", N=N, K=K, KC=KC, waves=waves, step_major=int(sm), z_shape=[KCz, wavesz, int(smz)], rows=a.rows,
                 exceptions_per_matrix=round(exc, 1), checked=bool(a.check))
        for tag, pk, nbytes in (("raw", wps, N * K * 2), ("z12", wzs, wzs[0].nbytes())):
            avg, _ = timed_graph(lambda i, pk=pk: run(pk, i), a.launches, lib)
            r[tag + "_us"] = round(avg * 1e3, 2)
            r[tag + "_MB"] = round(nbytes / 1e6, 2)
            r[tag + "_TBps"] = round(nbytes / 1e12 / (avg / 1e3), 3)
        r["speedup"] = round(r["raw_us"] / r["z12_us"], 3)
        print(json.dumps(r), flush=True)
        del wps, wzs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
