#!/usr/bin/env python3
"""Kernel overlap from a rocprofv3 --kernel-trace CSV: per kernel name its summed duration, and how
much of it ran while a kernel of ANOTHER queue was running."""
import csv, sys, collections, glob
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "0")))
rows.sort()
rows = rows[len(rows) // 3:]          # steady state
tot = collections.defaultdict(int); ov = collections.defaultdict(int)
for i, (s, e, n, q) in enumerate(rows):
    tot[n] += e - s
    for (s2, e2, n2, q2) in rows[max(0, i - 12): i + 12]:
        if q2 != q:
            ov[n] += max(0, min(e, e2) - max(s, s2))
span = rows[-1][1] - rows[0][0]
print("span %.3f ms, sum of kernel durations %.3f ms" % (span / 1e6, sum(tot.values()) / 1e6))
for n in sorted(tot, key=lambda k: -tot[k]):
    print("%-42s %8.3f ms  overlapped with the other queue %5.1f %%" % (n, tot[n] / 1e6, 100.0 * ov[n] / max(tot[n], 1)))
