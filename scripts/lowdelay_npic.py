#!/usr/bin/env python3
"""bench.lowdelay_8k by pictures per launch (GPU box): python3 scripts/lowdelay_npic.py
r03: 2 -> 0.189, 4 -> 0.177, 6 -> 0.177, 8 -> 0.177 ms per 8K picture (the DC prediction's launch is a latency
chain of constant length: 0.085 / 0.044 / 0.030 / 0.022 ms per picture)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import schroedinger_amd as sa

ctx = sa.Context(0)
for npic in (2, 4, 6, 8):
    r = bench.lowdelay_8k(ctx, npic=npic)
    print(npic, r["ms_per_picture"], r["kernels_ms_per_picture"], flush=True)
