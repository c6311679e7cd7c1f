#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel trace CSV by (kernel name, grid, workgroup): one kernel template serves several launch shapes
(the four G1 projections of a layer), which --stats merges.  usage: trace_by_grid.py <dir-or-csv> [min_calls]"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
paths = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
agg = collections.defaultdict(list)
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].replace("(anonymous namespace)", "anon").split("(")[0][-60:]
            grid = tuple(int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            wg = int(r.get("Workgroup_Size_X", 0) or 0)
            agg[(name, grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':60s} {'grid':>22s} {'wg':>5s} {'calls':>7s} {'avg_us':>9s} {'min_us':>8s} {'share':>6s}")
for (name, grid, wg), v in rows:
    if len(v) < min_calls:
        continue
    print(f"{name:60s} {str(grid):>22s} {wg:5d} {len(v):7d} {sum(v) / len(v) / 1e3:9.2f} {min(v) / 1e3:8.2f} {100 * sum(v) / tot:5.1f}%")
