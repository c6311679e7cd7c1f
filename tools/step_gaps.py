#!/usr/bin/env python3
"""Where a decode step is NOT running a kernel: from a rocprofv3 --kernel-trace CSV, per SJD iteration (k5_reguess to k5_reguess) the span, the
union of kernel intervals (all queues) and the largest gaps with the kernels on both sides.
usage: step_gaps.py <rocprofv3 output dir>"""
import collections
import csv
import glob
import json
import os
import sys

rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]))
rows.sort()
k5 = [i for i, r in enumerate(rows) if "k5_reguess" in r[2]]
steps = []
gaps = collections.Counter()
gap_ns = collections.defaultdict(int)
for a, b in zip(k5[-60:-1], k5[-59:]):
    seg = rows[a:b]
    span = rows[b][0] - seg[0][0]
    busy, cur_e = 0, seg[0][0]
    prev = None
    for s, e, n in seg:
        if s > cur_e and prev is not None:
            gaps[(prev, n)] += 1
            gap_ns[(prev, n)] += s - cur_e
        busy += max(0, e - max(s, cur_e))
        if e > cur_e:
            cur_e, prev = e, n
    tail = rows[b][0] - cur_e                     # from the last kernel of the step to the next step's first kernel: the host's turn
    steps.append((span, busy, tail))
n = len(steps)
print(json.dumps(dict(steps=n, span_us=round(sum(s[0] for s in steps) / n / 1e3, 1), kernels_running_us=round(sum(s[1] for s in steps) / n / 1e3, 1),
                      idle_us=round(sum(s[0] - s[1] for s in steps) / n / 1e3, 1), host_turn_us=round(sum(s[2] for s in steps) / n / 1e3, 1))))
for (a, b), ns in sorted(gap_ns.items(), key=lambda kv: -kv[1])[:12]:
    print(json.dumps(dict(after=a, before=b, per_step_us=round(ns / n / 1e3, 2), count_per_step=round(gaps[(a, b)] / n, 1))))
