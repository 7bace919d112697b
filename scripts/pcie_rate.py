#!/usr/bin/env python3
"""Pinned-memory copy rates of the box: H2D, D2H and both at once, through the library's copy queues."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import schroedinger_amd as sa

ctx = sa.Context(0)
n = 64 << 20
h_up, h_dn = ctx.host_array((1, n), np.uint8), ctx.host_array((1, n), np.uint8)
d_up, d_dn = ctx.plane(1, n, np.uint8), ctx.plane(1, n, np.uint8)
def run(up, dn, reps=8):
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if up:
            ctx.select_queue(ctx.QUEUE_H2D); d_up.upload_async(h_up)
        if dn:
            ctx.select_queue(ctx.QUEUE_D2H); d_dn.download_async(h_dn)
    ctx.select_queue(0); ctx.synchronize()
    return reps * n * (up + dn) / (time.perf_counter() - t0) / 1e9
run(1, 1, 2)
print("H2D %.1f GB/s  D2H %.1f GB/s  both %.1f GB/s (sum)" % (run(1, 0), run(0, 1), run(1, 1)))
