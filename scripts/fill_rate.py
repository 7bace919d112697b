import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import schroedinger_amd as sa
ctx = sa.Context(0)
for mb in (100, 200, 400):
    p = ctx.plane(mb * 1024, 1024, np.uint8)
    for _ in range(3): p.fill(1)
    ts = []
    for _ in range(10):
        ctx.timer_begin(); p.fill(2); ts.append(ctx.timer_end())
    print(mb, "MiB fill: median %.4f ms = %.2f TB/s" % (np.median(ts), mb * 1.048576e6 / np.median(ts) / 1e9))
    p.free()
