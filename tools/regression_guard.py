#!/usr/bin/env python3
"""Per-kernel regression guard (round 6): compares two `*_by_shape.txt` tables of tools/trace_by_grid.py -- the previous round's
committed one and the one just measured -- and FAILS (exit 1) when a hot kernel's average launch time moved by more than `--tol` (3 %).
A hot kernel = a (kernel, grid, workgroup) row with at least `--min-share` (1 %) of the old profile's time.  Kernel names are compared with
their template arguments, grids exactly: a launch shape that disappeared or appeared is reported, not failed (shapes change when kernels do).
  python tools/regression_guard.py profiles/r5final_bench_by_shape.txt gpurun_out/r6_bench_by_shape.txt
It would have caught round 5's k1_combine regression (5.2 -> 7.7 us, carried in by a refactor of the partial kernels) on the day it happened."""
import argparse
import re
import sys

ROW = re.compile(r"^(?P<name>.+?)\s+\((?P<grid>\d+, \d+, \d+)\)\s+(?P<wg>\d+)\s+(?P<calls>\d+)\s+(?P<avg>[\d.]+)\s+(?P<min>[\d.]+)\s+(?P<share>[\d.]+)%\s*$")


def table(path):
    rows = {}
    for ln in open(path):
        m = ROW.match(ln.rstrip("\n"))
        if m:
            rows[(m["name"].strip(), m["grid"], int(m["wg"]))] = (float(m["avg"]), float(m["min"]), float(m["share"]), int(m["calls"]))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("old")
    ap.add_argument("new")
    ap.add_argument("--tol", type=float, default=0.03)
    ap.add_argument("--min-share", type=float, default=1.0)
    ap.add_argument("--use-min", action="store_true", help="compare the minimum launch time instead of the average (less sensitive to a noisy box)")
    ap.add_argument("--absolute", action="store_true", help="do not take the median box drift out of the per-kernel ratios")
    a = ap.parse_args()
    old, new = table(a.old), table(a.new)
    if not old or not new:
        print("regression_guard: could not parse a table", file=sys.stderr)
        return 2
    bad = []
    # box-to-box drift: two boxes of the pool differ by 1-3 % on EVERY kernel (clock under the power cap; round 6's first run: all eleven hot
    # kernels +0.7..+3.4 %, median +2.0 %, bench.py's step time unchanged).  The median ratio over the hot kernels is reported and taken out of
    # each kernel's own ratio; a drift beyond the tolerance fails by itself (a uniform slowdown is still a slowdown).
    ratios = sorted((new[k][1] / v[1] if a.use_min else new[k][0] / v[0]) for k, v in old.items() if v[2] >= a.min_share and k in new)
    drift = (ratios[len(ratios) // 2] - 1.0) if (ratios and not a.absolute) else 0.0
    print(f"  box drift (median ratio over {len(ratios)} hot kernels): {100 * drift:+.1f} %")
    if drift > a.tol:
        bad.append(("<every kernel>", "", 0))
    for key, (avg, mn, share, calls) in sorted(old.items(), key=lambda kv: -kv[1][2]):
        if share < a.min_share:
            continue
        if key not in new:
            print(f"  gone      {key[0][-60:]:60s} {key[1]:>20s}  (was {avg:.2f} us, {share:.1f} %)")
            continue
        o, n = (mn, new[key][1]) if a.use_min else (avg, new[key][0])
        rel = n / o - 1.0 - drift
        flag = "SLOWER" if rel > a.tol else "faster" if rel < -a.tol else "ok"
        print(f"  {flag:9s} {key[0][-60:]:60s} {key[1]:>20s}  {o:8.2f} -> {n:8.2f} us  ({100 * (rel + drift):+.1f} %, {100 * rel:+.1f} % beyond the drift)")
        if rel > a.tol:
            bad.append(key)
    for key in new:
        if key not in old and new[key][2] >= a.min_share:
            print(f"  new       {key[0][-60:]:60s} {key[1]:>20s}  {new[key][0]:.2f} us, {new[key][2]:.1f} %")
    if bad:
        print(f"regression_guard: {len(bad)} hot kernel(s) slower than {100 * a.tol:.0f} % against {a.old}", file=sys.stderr)
        return 1
    print("regression_guard: no hot kernel moved by more than %.0f %%" % (100 * a.tol))
    return 0


if __name__ == "__main__":
    sys.exit(main())
