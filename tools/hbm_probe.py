#!/usr/bin/env python3
"""Calibration: what a plain streaming read / copy reaches on this MI355X box (torch ops, HIP events)."""
import json
import torch

dev = torch.device("cuda:0")
out = {}
for mb in (34, 100, 180, 1024):
    n = mb * 1024 * 1024 // 2
    xs = [torch.randn(n, device=dev, dtype=torch.bfloat16) for _ in range(max(2, 2048 // mb // 4))]
    y = torch.empty_like(xs[0])
    for name, fn, bytes_ in (("read_sum", lambda x: x.view(torch.int32).sum(), n * 2), ("copy", lambda x: y.copy_(x), n * 4)):
        for x in xs:
            fn(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 0
        e0.record()
        for r in range(4):
            for x in xs:
                fn(x)
                reps += 1
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[f"{name}_{mb}MB"] = dict(us=round(ms * 1e3, 2), TBps=round(bytes_ / 1e12 / (ms / 1e3), 3))
print(json.dumps(out))
