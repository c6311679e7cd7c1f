#!/usr/bin/env python3
"""Round 5, stage A of the loader / consumer engine: G1z as ONE persistent workgroup per CU with an LDS-DMA weight ring
(sjd_skinny_gemm_engine_z) against the product's g1z_skinny_gemm, per Lumina-7B projection shape, launches replayed from a hipGraph over distinct
weight sets (every launch streams from HBM).  us per launch and TB/s on the STORED bytes; results checked bit for bit first."""
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
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load_exp()
    import sjd_amd.backbones as BB
    prod = BB.ChameleonBackbone.G1_CFG_Z
    # (shape, N, K, engine KC, engine workgroups)
    cases = [("qkv", 12288, 4096, [(512, 256), (1024, 256)]), ("down", 4096, 11008, [(688, 256), (512, 242)]), ("o", 4096, 4096, [(512, 256), (1024, 128)])]
    x_of = {}
    for name, N, K, engs in cases:
        if a.only and name not in a.only.split(","):
            continue
        x = x_of.setdefault(K, torch.randn(32, K, device=dev).to(torch.bfloat16))
        ws = [(torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in range(a.copies)]
        KCp, wavesp, smp = prod[name]
        wzp = [ops.pack_weight_z(w, KCp, smp) for w in ws]
        t_prod = timed_graph(lambda i: ops.skinny_gemm(x, wzp[i % a.copies], N, K, KCp, wavesp, smp), a.launches, lib)[0] * 1e3
        mb = wzp[0].nbytes() / 1e6
        print(json.dumps(dict(shape=name, kernel="g1z_skinny_gemm (product)", cfg=[KCp, wavesp, int(smp)], stored_MB=round(mb, 1), us=round(t_prod, 2),
                              TBps_stored=round(mb / t_prod, 3))), flush=True)
        del wzp
        for KC, n_wg in engs:
            wze = [ops.pack_weight_z(w, KC, False) for w in ws]
            ref = ops.skinny_gemm(x, wze[0], N, K, KC, 8, False).data
            got = ops.skinny_gemm_engine(x, wze[0], n_wg=n_wg).data
            torch.cuda.synchronize()
            same = bool(torch.equal(ref.view(torch.int32), got.view(torch.int32)))
            t_same = timed_graph(lambda i: ops.skinny_gemm(x, wze[i % a.copies], N, K, KC, 8, False), a.launches, lib)[0] * 1e3
            t_eng = timed_graph(lambda i: ops.skinny_gemm_engine(x, wze[i % a.copies], n_wg=n_wg), a.launches, lib)[0] * 1e3
            mbe = wze[0].nbytes() / 1e6
            print(json.dumps(dict(shape=name, kernel="g1e_skinny_gemm (loader / consumer)", KC=KC, workgroups=n_wg, bit_identical=same, stored_MB=round(mbe, 1),
                                  engine_us=round(t_eng, 2), engine_TBps_stored=round(mbe / t_eng, 3), g1z_same_packing_8waves_us=round(t_same, 2),
                                  product_us=round(t_prod, 2), timeouts=ops.engine_timeouts())), flush=True)
            del wze
        del ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
