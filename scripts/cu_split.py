#!/usr/bin/env python3
"""CU partitioning between the two kernel queues: the HBM-bound stages (upsample +
inverse wavelet) on queue 0 restricted to some compute units, OBMC on queue 1 restricted to the others.
python3 scripts/cu_split.py   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import schroedinger_amd as sa

NCU = 256


def run(label, pipe, order, masks):
    os.environ["SCHRO_BENCH_PIPE"] = pipe
    os.environ["SCHRO_BENCH_ORDER"] = str(order)
    ctx = sa.Context(0)
    if masks:
        ctx.queue_set_cu_mask(0, masks[0])
        ctx.queue_set_cu_mask(1, masks[1])
    wl = bench.Workload(ctx, 8, seed=1, queues=2)
    for _ in range(10):
        wl.step()
    ctx.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        wl.step()
    ctx.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    print("%-58s %.4f ms per step" % (label, ms), flush=True)
    ctx.close() if hasattr(ctx, "close") else None


run("whole batches per queue, no masks (the bench's default)", "whole", 2, None)
run("split: queue 0 upsample + wavelet, queue 1 OBMC, no masks", "split", 0, None)
for period, hbm in [(4, 1), (8, 3), (2, 1), (8, 1), (16, 3)]:
    q0 = [1 if (i % period) < hbm else 0 for i in range(NCU)]
    q1 = [1 - b for b in q0]
    run("split, CU i -> queue 0 if i %% %d < %d (%d CUs), else queue 1" % (period, hbm, sum(q0)), "split", 0, (q0, q1))
for first in (64, 96):
    q0 = [1 if i < first else 0 for i in range(NCU)]
    q1 = [1 - b for b in q0]
    run("split, CUs 0 .. %d -> queue 0, the rest queue 1" % (first - 1), "split", 0, (q0, q1))
q0 = [1] * NCU
q1 = [0 if (i % 4) == 0 else 1 for i in range(NCU)]
run("split, queue 0 everywhere, queue 1 without every 4th CU", "split", 0, (q0, q1))
