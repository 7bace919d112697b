#!/usr/bin/env python3
"""Config 5 (BASELINE.json configs[4]) on one MI355X: VC-2 low-delay 10-bit 4:2:2 7680x4320,
s32 coefficients, 3-level Haar (no shift).  Times slice decode, DC prediction and the
inverse wavelet per picture with the library's own HIP events, checks the coefficient
planes against the CPU oracle and times the oracle on the host for the same picture.
Prints one JSON line.  (bench.py measures the metric of record, config 4.)  Lives under tests/
because it needs the oracle as input generator and checker:
`python3 tests/bench_lowdelay.py [pictures per launch] [steps] [queues]`.
With two queues (default) two picture batches alternate between the context's in-order queues,
so the DC prediction of one batch (a dependency chain on a few CUs) runs beside the slice decode
or the wavelet of the other; the per-kernel figures come from a one-queue pass."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))      # (this directory)
import oracle_lib as O  # noqa: E402  (checker + CPU number only)
import schroedinger_amd as sa  # noqa: E402
import synth  # noqa: E402

NPIC = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
QUEUES = int(sys.argv[3]) if len(sys.argv) > 3 else 2
W, H, DEPTH, FILT = 7680, 4320, 3, 3


def main():
    ctx = sa.Context(0)
    P = synth.lowdelay_params(W, H, (1, 0), DEPTH, 32, 8, 155, 1)
    rng = np.random.default_rng(5)
    dims = [(P["iwt_luma_height"], P["iwt_luma_width"])] + [(P["iwt_chroma_height"], P["iwt_chroma_width"])] * 2
    q = []
    for (h, w) in dims:         # Laplacian-like quantised values, ~2.1 bits per sample
        mag = np.floor(rng.exponential(0.9, (h, w))).astype(np.int32)
        q.append(np.where(rng.integers(0, 2, (h, w)) == 1, -mag, mag).astype(np.int32))
    bi = rng.integers(4, 29, P["n_horiz_slices"] * P["n_vert_slices"]).astype(np.uint8)
    data = O.lowdelay_write(q, P, 4, bi)
    del q
    want = [np.zeros(d, np.int32) for d in dims]
    t0 = time.perf_counter()
    O.lowdelay_decode(data, want, P)
    cpu_s = time.perf_counter() - t0

    def batch():
        pics = []
        for _ in range(NPIC):
            sl = ctx.upload_bytes(data)
            co = [ctx.plane(h, w, np.int32) for (h, w) in dims]
            px = [ctx.plane(h, w, np.int32) for (h, w) in dims]
            pics.append((sl, co, px))
        return pics, [(sl, co) for sl, co, _ in pics], [(c, p) for _, co, px in pics for c, p in zip(co, px)]
    sets = [batch() for _ in range(QUEUES)]
    pics, jobs, pairs = sets[0]
    ok = True
    for q, (pics_q, jobs_q, _) in enumerate(sets):
        ctx.select_queue(q)
        ctx.lowdelay_batch(jobs_q, P)
        ctx.synchronize()
        ok = ok and all(np.array_equal(pics_q[-1][1][k].download(), want[k]) for k in range(3))
    ctx.select_queue(0)

    # per-kernel times: one batch after the other on one queue
    for _ in range(2):
        ctx.lowdelay_batch(jobs, P)
        ctx.iiwt_batch(pairs, DEPTH, FILT)
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_reset()
    ctx.timer_begin()
    for _ in range(STEPS):
        ctx.lowdelay_batch(jobs, P)
        ctx.iiwt_batch(pairs, DEPTH, FILT)
    wall1 = ctx.timer_end() / STEPS
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    # the figure of record: batches alternating between the queues
    def step(k):
        _, jobs_q, pairs_q = sets[k % QUEUES]
        ctx.select_queue(k % QUEUES)
        ctx.lowdelay_batch(jobs_q, P)
        ctx.iiwt_batch(pairs_q, DEPTH, FILT)
    for k in range(2 * QUEUES):
        step(k)
    ctx.synchronize()
    t0 = time.perf_counter()
    for k in range(2 * STEPS):
        step(k)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / (2 * STEPS)
    ctx.select_queue(0)
    samples = sum(h * w for h, w in dims)
    per = {k: ms / n for k, (ms, n) in prof.items() if n}
    iiwt = sum(ms for k, (ms, n) in prof.items() if k.startswith("iiwt")) / STEPS
    print(json.dumps({
        "workload": "7680x4320 4:2:2 s32 low-delay, 32x8 slices of 155 bytes, depth 3 Haar, %d pictures per launch, "
                    "%d batch(es) in flight" % (NPIC, QUEUES),
        "parity_vs_oracle": "bit-exact" if ok else "MISMATCH",
        "ms_per_picture": {"slices": per.get("slices", 0) / NPIC, "dc_predict": per.get("dc_predict", 0) / NPIC,
                           "iiwt": iiwt / NPIC, "wall_one_queue": wall1 / NPIC, "wall": wall / NPIC},
        "Mpix_per_s": W * H * NPIC / wall / 1e3,
        "slice_GBps_written": samples * 4 * NPIC / per.get("slices", 1) / 1e6,
        "compressed_MB_per_picture": data.size / 1e6, "coefficient_MB_per_picture": samples * 4 / 1e6,
        "cpu_oracle_ms_per_picture": cpu_s * 1e3, "cpu_threads": 1}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
