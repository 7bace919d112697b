import os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
import schroedinger_amd as sa
ctx = sa.Context(0)
wl = bench.Workload(ctx, 8, seed=1, queues=2)
c, b = ctx, wl.sets[0]
hand = []
for f in range(wl.frames):
    for k, (h, w) in enumerate(wl.dims):
        dst = b.iwt_pairs[3 * f + k][0]
        blob, cbs = bench.quantised_handover(h, w, bench.DEPTH, dst.stride, 700 + 3 * f + k)
        hand.append((dst, c.upload(blob.reshape(1, -1)), c.codeblock_table(cbs), blob.size))
print("codeblocks", sum(len(t) for _, _, t, _ in hand))
jobs = [(d, v, t, False) for d, v, t, _ in hand]
for _ in range(3):
    c.dequant_batch(jobs, 0)
c.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        c.dequant_batch(jobs, 0)
    t1 = time.perf_counter()
    c.synchronize()
    t2 = time.perf_counter()
    print("host %.3f ms per call, with drain %.3f" % ((t1 - t0) * 100, (t2 - t0) * 100))
