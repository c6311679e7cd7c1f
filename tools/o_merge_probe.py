#!/usr/bin/env python3
"""Gate probe for VERDICT r5 next #4 ("the o projection consumes K1's split partials"): the staging prologue such an o projection would run --
every workgroup of its (22 x 8) grid merging the 4 split triples of its chunk's 4 heads for 32 window rows, 266 KB of L2-resident partials per
workgroup instead of 32 KB of merged activations -- timed alone in a hipGraph against the same launch with an empty body
(csrc/sjd_gemm.hip::o_merge_prologue_probe, libsjd_hip_exp.so).  What it must beat: k1_combine (4.98 us in the decode's trace) + one kernel
boundary, minus the 32-KB staging it replaces; the gate of the verdict: k1_partial -> o pair <= 20.5 us, i.e. the o projection may grow by
at most ~0.9 us over its 8.7 us."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sjd_amd._lib as L  # noqa: E402


def main():
    lib = L.load_exp()
    dev = "cuda:0"
    rows, n_split = 32, 4
    part = torch.randn(32, n_split, rows, 130, device=dev)
    part[..., 1] = part[..., 1].abs() + 1.0
    sink = torch.zeros(64, device=dev)
    st = torch.cuda.Stream()
    res = {}
    with torch.cuda.stream(st):
        for mode in (0, 1):
            def launch():
                for _ in range(32):
                    L.check(lib.sjd_o_merge_prologue_probe(part.data_ptr(), sink.data_ptr(), n_split, rows, mode, ctypes.c_void_p(st.cuda_stream)), "probe")
            launch()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                launch()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(20):
                g.replay()
            e1.record(st)
            torch.cuda.synchronize()
            res[mode] = e0.elapsed_time(e1) * 1e3 / (20 * 32)
    out = {"probe": "o-projection staging prologue that merges K1's split partials (22 x 8 workgroups, 32 rows x 4 heads x 4 splits each)",
           "empty_launch_us": round(res[0], 2), "merge_prologue_launch_us": round(res[1], 2), "prologue_us": round(res[1] - res[0], 2),
           "bytes_per_workgroup": rows * 4 * n_split * 130 * 4, "bytes_per_launch": 176 * rows * 4 * n_split * 130 * 4,
           "budget_us": "k1_combine 4.98 + one kernel boundary ~1.5 - the 32-KB activation staging it replaces ~0.5; verdict gate: o projection + <= 0.9"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
