import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, bench, schroedinger_amd as sa
ctx = sa.Context(0)
c = ctx
wl = bench.Workload(ctx, 8, seed=1, queues=2)
hs = [bench.HostSide(wl, b, True, 900 + 100 * i) for i, b in enumerate(wl.sets)]
T = [0.0] * 6
def step(k):
    i = k % 2
    b, h = wl.sets[i], hs[i]
    t = [time.perf_counter()]
    c.select_queue(c.QUEUE_H2D); c.queue_wait_mark(8 + i)
    h.d_blob.block.upload_async(h.blob); b.mv_arena.block.upload_async(h.mv); c.queue_mark(i)
    t.append(time.perf_counter())
    c.select_queue(k % 2); c.queue_wait_mark(i); c.queue_wait_mark(12 + i)
    t.append(time.perf_counter())
    c.dequant_batch([(dst, dev, tab, False) for dst, dev, tab in h.hand], 0)
    t.append(time.perf_counter())
    c.upsample_batch(b.up_pairs); c.iiwt_batch(b.iwt_pairs, 3, 0); c.obmc_batch(b.obmc_jobs)
    c.queue_mark(8 + i); c.queue_mark(4 + i)
    t.append(time.perf_counter())
    c.select_queue(c.QUEUE_D2H); c.queue_wait_mark(4 + i)
    b.out_arena.block.download_async(h.out); c.queue_mark(12 + i)
    t.append(time.perf_counter())
    for j in range(5): T[j] += t[j + 1] - t[j]
for k in range(4): step(k)
c.select_queue(0); c.synchronize()
T = [0.0] * 6
t0 = time.perf_counter()
for k in range(4, 16): step(k)
t1 = time.perf_counter()
c.select_queue(0); c.synchronize()
print("per step ms: h2d %.3f waits %.3f dequant %.3f kernels %.3f d2h %.3f; enqueue %.3f total %.3f" % tuple([x / 12 * 1e3 for x in T[:5]] + [(t1 - t0) / 12 * 1e3, (time.perf_counter() - t0) / 12 * 1e3]))
