#!/usr/bin/env python3
"""Host time of bench.py's step: how long the Python thread takes to enqueue a step (no waiting), against the
device's time per step.  If the two are close, the headline is bound by the host loop, not by the kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import schroedinger_amd as sa

ctx = sa.Context(0)
wl = bench.Workload(ctx, 8, seed=1, queues=2)
for _ in range(50):
    wl.step()
ctx.synchronize()
n = 400
t0 = time.perf_counter()
for _ in range(n):
    wl.step()
t1 = time.perf_counter()
ctx.synchronize()
t2 = time.perf_counter()
print("host enqueue %.4f ms per step; device done after %.4f ms per step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
