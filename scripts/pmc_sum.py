#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection CSVs per (kernel, counter), per launch."""
import csv, glob, re, sys, collections

def short(n):
    m = re.search(r"(\w+<[^(]*>)\(", n) or re.search(r"(\w+)\(", n)
    return (m.group(1) if m else n)[-60:]

for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            key = (k, r["Counter_Name"])
            acc[key][0] += float(r["Counter_Value"])
            acc[key][1] = max(acc[key][1], int(r["Dispatch_Id"]))
    disp = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            disp[short(r["Kernel_Name"])].add(r["Dispatch_Id"])
    for (k, c), (v, _) in sorted(acc.items()):
        n = len(disp[k])
        print("%-62s %-28s %14.0f per launch (%d launches)" % (k, c, v / n, n))
