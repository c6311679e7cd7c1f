#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs per kernel: average counter value per dispatch.
usage: pmc_summary.py <dir> [kernel-substring ...]   -> one JSON line per (kernel, counter)
Used for profiles/*_traffic.json (FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 on gfx950 for wide coalesced streams, see
MI355X_MICROARCH.md) and profiles/*_mfma_busy.json (SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE)."""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
filters = sys.argv[2:]
paths = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0]
            if filters and not any(s in name for s in filters):
                continue
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r.get("End_Timestamp") and r.get("Start_Timestamp"):
                dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for name, ctrs in agg.items():
    row = {"kernel": name[-90:], "dispatches": max(len(v) for v in ctrs.values())}
    for c, v in sorted(ctrs.items()):
        row[c] = round(sum(v) / len(v), 2)
    if dur[name]:
        row["avg_us_under_pmc"] = round(sum(dur[name]) / len(dur[name]) / 1e3, 2)
    print(json.dumps(row))
