#!/usr/bin/env python3
"""rocprofv3 kernel trace -> per (kernel, grid size) durations of the dispatches that ran ALONE (no other dispatch of the
trace overlaps them): count, median, min, max in microseconds.  The stats CSV averages launches of one kernel name whatever
their grid (the wavelet's levels are one kernel) and whatever ran beside them (two batches in flight)."""
import csv
import re
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r) for r in rows)
groups = {}
for i, (a, b, r) in enumerate(ev):
    alone = (i == 0 or ev[i - 1][1] <= a) and (i + 1 == len(ev) or ev[i + 1][0] >= b)
    name = r["Kernel_Name"]
    m = re.search(r"(\w+<[^(]*>)\(", name) or re.search(r"(\w+)\(", name)
    key = ((m.group(1) if m else name)[-48:], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    groups.setdefault(key, [[], 0])
    groups[key][1] += 1
    if alone:
        groups[key][0].append((b - a) / 1e3)
print("%-50s %10s %7s %7s %9s %9s %9s" % ("kernel", "workgroups", "calls", "alone", "median us", "min us", "max us"))
for (k, g), (d, n) in sorted(groups.items()):
    if d:
        print("%-50s %10d %7d %7d %9.2f %9.2f %9.2f" % (k, g, n, len(d), statistics.median(d), min(d), max(d)))
    else:
        print("%-50s %10d %7d %7d" % (k, g, n, 0))
