#!/usr/bin/env python3
"""OBMC alone on bench.py's workload (GPU box): wall time of a luma-only and a chroma-only batch
over 20 launches, and -- with SCHRO_HIP_OBMC_STAMPS=1 -- the row kernel's in-kernel phase stamps
(cycles since the workgroup started: 7 the job is here, 10 accumulator cleared / ramps / vectors asked for, 8 the barrier
behind them, 1 set-up done (weight tables), 2 decode, 3 items, 4 passes, 5 rim, 6 barrier, 9 end) and the residency per CU.
r04, luma, prediction-only jobs (medians of 16 320 tiles): 808 / 3280 / 3628 / 5728 / 9564 / 12508 / 22512 / 22624 / 22872 /
23652 -- with 6.4 workgroups of four waves per CU every vector instruction of a wave waits for six others': the phases'
lengths are their instruction counts.
PLANES=luma|chroma restricts the run; MOTION=random|smooth|const replaces the bench's motion
field (uniform in +-16 pel per block) by a pan + slow zoom or by one vector per reference:
the launch takes the same time with all three."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import bench  # noqa: E402
import schroedinger_amd as sa  # noqa: E402
import synth  # noqa: E402


def main():
    ctx = sa.Context(0)
    wl = bench.Workload(ctx, 8, seed=1, queues=1)
    wl.step()
    ctx.synchronize()
    b = wl.sets[0]
    nbx, nby = wl.P["x_num_blocks"], wl.P["y_num_blocks"]
    kind = os.environ.get("MOTION", "random")
    # (r04: the prediction-only jobs of the combine form, as bench.py's step runs them; FORM=add: the r03 jobs)
    jobs = b.obmc_jobs if os.environ.get("FORM") == "add" or not b.pred_jobs else b.pred_jobs
    if kind != "random":
        jobs = []
        for f in range(wl.frames):
            mv = b.mv_np[f].copy()
            mode = mv["flags"] & 3
            if kind == "const":
                vec = np.tile(np.array([5, -7, 3, 9], np.int16), (mv.shape[0], 1))
            elif kind == "zero":        # every window right behind its block, one tap: the most line sharing there is
                vec = np.zeros((mv.shape[0], 4), np.int16)
            elif kind == "y2":          # ... two rows down: the 12 rows of a window start on a band of 4 plane rows (zero: in the middle of one)
                vec = np.tile(np.array([0, 8, 0, 8], np.int16), (mv.shape[0], 1))
            elif kind == "q11":         # all four quarter-pel taps of windows right behind their blocks
                vec = np.tile(np.array([1, 1, 1, 1], np.int16), (mv.shape[0], 1))
            elif kind == "q19":         # ... and two rows down (band-aligned windows)
                vec = np.tile(np.array([1, 9, 1, 9], np.int16), (mv.shape[0], 1))
            elif kind == "even":        # the bench's independent vectors rounded to half-pel positions of even parity: one tap each
                vec = (mv["v"] & ~np.int16(3)).astype(np.int16)
            elif kind == "odd":         # ... and with every tap needed (all four quarter-pel taps of both references)
                vec = (mv["v"] | np.int16(1)).astype(np.int16)
            else:           # a pan + a slow zoom, +-1 quarter pel of noise
                yy, xx = np.divmod(np.arange(nbx * nby), nbx)
                n = synth.lcg(4 * nbx * nby, 77 + f).reshape(4, -1) % 3 - 1
                vec = np.stack([5 + xx // 64 + n[0], -7 + xx // 48 + n[1], 3 + yy // 64 + n[2],
                                9 - yy // 48 + n[3]], 1).astype(np.int16)
            mv["v"] = np.where((mode == 0)[:, None], mv["v"], vec)
            d_mv = ctx.upload_bytes(mv)
            for k in range(3):
                jobs.append(sa.obmc_plane(d_mv, wl.P, k, b.hp[0][0][k], b.hp[0][1][k], None, b.iwt_combine[3 * f + k][2], prediction_only=True))
    which = os.environ.get("PLANES", "luma,chroma").split(",")
    sets = (("luma", [j for i, j in enumerate(jobs) if i % 3 == 0]), ("chroma", [j for i, j in enumerate(jobs) if i % 3]))
    for name, sel in sets:
        if name not in which:
            continue
        for _ in range(3):
            ctx.obmc_batch(sel)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.obmc_batch(sel)
        ctx.synchronize()
        print(kind, name, "%.4f ms" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
    sa._lib.load().schro_hip_obmc_stamps_dump()


if __name__ == "__main__":
    main()
