#!/usr/bin/env python3
"""Where the host's time goes in bench.pcie_pipeline's step (quantised hand-over): per call, ms per step."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
import schroedinger_amd as sa

ctx = sa.Context(0)
wl = bench.Workload(ctx, 8, seed=1, queues=2)
for _ in range(3):
    wl.step()
c = ctx
c.select_queue(0); c.synchronize()
wl.pcie_sets = list(wl.sets) + [bench.BatchSet(wl, 4242)]
sets = wl.pcie_sets
nb = len(sets)
hs = [bench.HostSide(wl, b, True, 900 + 100 * i) for i, b in enumerate(sets)]
acc = collections.defaultdict(float)

def timed(name, fn, *a):
    t0 = time.perf_counter()
    r = fn(*a)
    acc[name] += time.perf_counter() - t0
    return r

def step(k):
    i = k % nb
    b, h = sets[i], hs[i]
    c.select_queue(c.QUEUE_H2D)
    timed("wait_mark", c.queue_wait_mark, 8 + i)
    timed("upload blob", h.d_blob.block.upload_async, h.blob)
    timed("upload mv", b.mv_arena.block.upload_async, h.mv)
    timed("mark", c.queue_mark, i)
    c.select_queue(k % 2)
    timed("wait_mark", c.queue_wait_mark, i)
    timed("wait_mark", c.queue_wait_mark, 12 + i)
    if os.environ.get("SKIP_DEQUANT") == "1":
        pass
    elif os.environ.get("SCHRO_BENCH_DEQUANT_PLAN", "1") != "0":
        timed("dequant (plan)", lambda: wl.dq_plan.run(planes=h.dq_planes))
    else:
        timed("dequant", c.dequant_batch, [(dst, dev, tab, False) for dst, dev, tab in h.hand], 0)
    timed("upsample", c.upsample_batch, b.up_pairs)
    timed("iiwt", c.iiwt_batch, b.iwt_pairs, bench.DEPTH, bench.FILTER)
    timed("obmc", c.obmc_batch, b.obmc_jobs)
    timed("mark", c.queue_mark, 8 + i)
    timed("mark", c.queue_mark, 4 + i)
    c.select_queue(c.QUEUE_D2H)
    timed("wait_mark", c.queue_wait_mark, 4 + i)
    timed("download", b.out_arena.block.download_async, h.out)
    timed("mark", c.queue_mark, 12 + i)

for k in range(4):
    step(k)
c.select_queue(0); c.synchronize()
acc.clear()
t0 = time.perf_counter()
n = 12
for k in range(4, 4 + n):
    step(k)
th = time.perf_counter() - t0
c.select_queue(0); c.synchronize()
print("host %.3f ms per step, with drain %.3f" % (th / n * 1e3, (time.perf_counter() - t0) / n * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-12s %.3f ms per step" % (k, v / n * 1e3))
