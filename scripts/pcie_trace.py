#!/usr/bin/env python3
"""Only the PCIe-inclusive pipelines of bench.py (for rocprofv3 --kernel-trace --memory-copy-trace):
python3 scripts/pcie_trace.py [dense|quantised] [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import schroedinger_amd as sa

kind = sys.argv[1] if len(sys.argv) > 1 else "quantised"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = sa.Context(0)
wl = bench.Workload(ctx, 8, seed=1, queues=2)
for _ in range(3):
    wl.step()
print(json.dumps(bench.pcie_pipeline(wl, quantised=(kind == "quantised"), steps=steps)))
