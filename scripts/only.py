#!/usr/bin/env python3
"""One of bench.py's extra legs by itself, for profiler runs:
python3 scripts/only.py iiwt_1080p [frames]|iiwt_2160p|iiwt_s32_2160p|lowdelay_8k|pcie_dense|pcie_quantised"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import schroedinger_amd as sa

what = sys.argv[1]
ctx = sa.Context(0)
if what == "iiwt_1080p":
    print(json.dumps(bench.iiwt_1080p(ctx, frames=int(sys.argv[2]) if len(sys.argv) > 2 else 8)))
elif what == "iiwt_s32_2160p":
    print(json.dumps(bench.iiwt_s32_2160p(0)))
elif what == "iiwt_2160p":      # the plain 3-level DD(9,7) transform of 8 x 2160p (north_star's own target), its launches alone
    wl = bench.Workload(ctx, 8, seed=1, queues=2)
    print(json.dumps(bench.iiwt_2160p(wl)))
elif what == "lowdelay_8k":
    print(json.dumps(bench.lowdelay_8k(ctx)))
else:
    wl = bench.Workload(ctx, 8, seed=1, queues=2)
    for _ in range(3):
        wl.step()
    print(json.dumps(bench.pcie_pipeline(wl, quantised=(what == "pcie_quantised"), form=(sys.argv[2] if len(sys.argv) > 2 else None))))
