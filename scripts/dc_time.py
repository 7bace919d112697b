#!/usr/bin/env python3
"""DC prediction alone, by band shape: python3 scripts/dc_time.py  (per-launch HIP-event time)
Shapes separate the cost of a step (one strip of 64 rows) from the lag between strips."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import schroedinger_amd as sa

ctx = sa.Context(0)
SHAPES = [(960, 64, 1, np.int32), (960, 64, 12, np.int32), (1920, 64, 1, np.int32), (960, 128, 1, np.int32),
          (960, 540, 1, np.int32), (960, 540, 12, np.int32), (480, 540, 12, np.int32), (960, 540, 12, np.int16)]
if len(sys.argv) > 1:           # w h n
    SHAPES = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), np.int32)]
for (w, h, n, dt) in SHAPES:
    # planes as the LL band of a depth-3 frame: rows 8 frame rows apart
    planes = []
    for _ in range(n):
        full = ctx.plane(h * 8, w * 8, dt).fill(1)
        class V: pass
        v = V()
        v.ptr, v.stride, v.width, v.height, v.dtype = full.ptr, full.stride * 8, w, h, np.dtype(dt)
        planes.append((full, v))
    views = [v for _, v in planes]
    for _ in range(3):
        ctx.dc_predict_batch(views)
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_reset()
    reps = 10
    for _ in range(reps):
        ctx.dc_predict_batch(views)
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    ms = prof["dc_predict"][0] / reps
    steps = w + h
    print("%4d x %3d x %2d %-5s  %.4f ms per launch   %.1f ns per (w + h) step" % (w, h, n, np.dtype(dt).name, ms, ms * 1e6 / steps))
    for full, _ in planes:
        full.free()
    sys.stdout.flush()
