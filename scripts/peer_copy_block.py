#!/usr/bin/env python3
"""How long the exec-domain thread of the device that NEEDS a foreign reference sits inside frame_copy_to_async's
hipMemcpyPeerAsync while the producer's `ready` event has not fired (ROCm 7.2: an asynchronous copy behind an unfired
cross-queue wait returns only when the event has).  Two exec-domain threads and contexts
on device 0 (schro_hip_scheduler_new_on ({0, 0})); the reference picture's function enqueues PILE x 2 upsamples of a 2160p
plane (~30 us each) in front of its own upsample and returns at once; the dependent, on the other context, is submitted
right away.  Run under  rocprofv3 --hip-trace --output-format csv -d DIR -o run -- python3 scripts/peer_copy_block.py
and read the hipMemcpyPeerAsync rows of DIR/**/run_hip_api_trace.csv (scripts/peer_copy_block.py --parse DIR does);
without a profiler it prints the wall times it can see itself."""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def parse(d):
    import csv
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "MemcpyPeerAsync" in r.get("Function", ""):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Thread_Id", "")))
    rows.sort()
    for s, e, t in rows:
        print("hipMemcpyPeerAsync thread %s: %.3f ms inside the call" % (t, (e - s) * 1e-6))
    if rows:
        print("longest: %.3f ms, all: %.3f ms over %d calls" % (max(e - s for s, e, _ in rows) * 1e-6, sum(e - s for s, e, _ in rows) * 1e-6, len(rows)))


def main():
    import numpy as np
    import schroedinger_amd as sa
    import synth
    from schroedinger_amd import _lib, frames
    pile = int(os.environ.get("PILE", "60"))
    W, H = 320, 192
    sched = sa.Scheduler(devices=[0, 0])
    lib = sched.lib
    fmt8 = frames.frame_format(np.uint8, 1, 1)
    dims = [(H, W), (H // 2, W // 2), (H // 2, W // 2)]
    refs_np = [synth.picture_u8(h, w, seed=3 + k) for k, (h, w) in enumerate(dims)]
    c0 = sched.contexts[0]
    big = (c0.upload(synth.picture_u8(2160, 3840, seed=9)), [c0.hp_plane(2160, 3840) for _ in range(2)])
    plain = frames.DeviceFrame(c0, fmt8, W, H).upload(frames.HostFrame(refs_np, 1, 1))
    up = frames.DeviceFrame(c0, fmt8, W, H, upsampled=True)
    c0.synchronize()
    times = {}

    def reference(ctx, dev):
        t0 = time.perf_counter()
        src, hps = big
        for _ in range(pile):
            for hp in hps:
                ctx.upsample_batch([(src, hp)])
        c, pc = up.c.components, plain.c.components
        planes = (_lib.UpsamplePlane * 2)()
        planes[0] = _lib.UpsamplePlane(pc[0].data, pc[0].stride, c[0].data, c[0].stride, pc[0].width, pc[0].height, None, 0)
        planes[1] = _lib.UpsamplePlane(pc[1].data, pc[1].stride, c[1].data, c[1].stride, pc[1].width, pc[1].height, pc[2].data, pc[2].stride)
        sa.check(lib.schro_hip_upsample_batch(ctx.h, planes, 2))
        up.c.upsample_done = 1
        sched.publish_reference(dev, up.ptr())
        times["producer_enqueue_ms"] = (time.perf_counter() - t0) * 1e3
        times["producer_returned"] = time.perf_counter()
        return 0

    def dependent(ctx, dev):
        # (the scheduler has moved the frame by the time the function runs: what is measured is how long after the
        # producer's function returned that was)
        times["dependent_started_after_ms"] = (time.perf_counter() - times["producer_returned"]) * 1e3
        assert sched.reference_frame(dev, 0)
        return 0

    t0 = time.perf_counter()
    d0, _ = sched.submit(0, [], True, reference)
    d1, _ = sched.submit(10, [], True, lambda ctx, dev: 0)      # (puts the second chain on the other context)
    dev, foreign = sched.submit(11, [10, 0], False, dependent)
    assert dev == d1 and foreign == 0
    assert sched.wait() == 0
    c0.synchronize()
    times["all_ms"] = (time.perf_counter() - t0) * 1e3
    times.pop("producer_returned")
    print({k: round(v, 3) for k, v in times.items()}, "pile", pile)
    sched.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        main()
