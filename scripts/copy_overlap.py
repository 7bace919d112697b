#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace --memory-copy-trace run: busy time of kernels, H2D and D2H copies
over the steady-state span, and how much of the copy time ran while a kernel was running."""
import csv, glob, sys
def load(pat, name_key):
    rows = []
    for f in glob.glob(sys.argv[1] + "/**/*" + pat, recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get(name_key, "")))
    return sorted(rows)
k = load("kernel_trace.csv", "Kernel_Name")
c = load("memory_copy_trace.csv", "Direction")
t0 = k[len(k) // 3][0]; t1 = k[-1][1]
def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    return tot + (cur_e - cur_s if cur_e is not None else 0)
clip = lambda rows, pred=lambda n: True: [(max(s, t0), min(e, t1)) for s, e, n in rows if e > t0 and s < t1 and pred(n)]
ki = clip(k)
span = t1 - t0
print("steady-state span %.2f ms; kernels busy %.2f ms" % (span / 1e6, union(ki) / 1e6))
for d in sorted(set(n for _, _, n in c)):
    ci = clip(c, lambda n, d=d: n == d)
    big = [(s, e) for s, e in ci if e - s > 20000]
    both = union(ki + big) 
    print("%-28s %4d copies, busy %.2f ms (%.0f %% of the span); with kernels: union %.2f ms => %.2f ms of it beside kernels"
          % (d, len(ci), union(ci) / 1e6, 100.0 * union(ci) / span, both / 1e6, (union(ki) + union(big) - both) / 1e6))
