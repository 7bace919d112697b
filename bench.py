#!/usr/bin/env python3
"""bench.py -- Mpix/s of the Dirac/VC-2 decode pixel path on MI355X.

Workload (BASELINE.json configs[3], the 2160p configuration the metric is
quoted on; it fits one GPU): synthetic 3840x2160 4:2:0 inter pictures,
3-level Deslauriers-Dubuc (9,7) inverse wavelet on s16 coefficients (Y,U,V),
12x12/8x8 OBMC at quarter-pel from two references with the residual add and
u8 clamp fused, plus the half-pel upsampling of both references.

One "step" = one batch of --frames pictures through that path, everything
resident in HBM.  Frames shard across GPUs (one process per GPU, no data-path
collective): weak scaling, value = pictures * 3840*2160 / max-over-ranks time.

Prints ONE JSON line (rank 0).  The roofline block is the dominant kernel's
algorithmic bytes / its HIP-event time, measured during the timed steps; the
cpu_baseline block is the CPU oracle (a port of the reference algorithm, see
oracle/) timed on a bounded sample of the same workload, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import synth  # noqa: E402  (synthetic input generators shared with tests/)

W, H, DEPTH, FILTER = 3840, 2160, 3, 0
XBLEN, XBSEP, PREC = 12, 8, 2
REF_GROUP = 8                   # pictures that share a pair of references
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec


def coeff_plane(h, w, seed):
    """Synthetic s16 coefficients in the reference's in-place sub-band layout:
    small values in the detail bands, larger ones in the depth-3 LL band."""
    v = synth.lcg(h * w, seed).astype(np.int32).reshape(h, w)
    c = (v & 0x3f) - 32
    ll = ((v >> 6) & 0x3ff) - 512
    c[0::8, : w // 8] = ll[0::8, : w // 8]
    return c.astype(np.int16)


class BatchSet:
    """One picture batch in flight: its references' half-pel images, coefficient frames,
    motion fields, residual frames and output pictures -- all resident in HBM.  Coefficient frames,
    motion fields and output pictures each live in ONE device block per batch (an arena), so that a
    batch crosses the host boundary as one copy per kind (the PCIe-inclusive figures)."""

    def __init__(self, wl, seed):
        import schroedinger_amd as sa
        ctx, dims = wl.ctx, wl.dims
        # a pair of references per group of 8 pictures (wl.groups), upsampled once per batch: the luma planes'
        # half-pel image and ONE (U, V) pair image for the chroma planes (include/schro_hip.h, r04;
        # SCHRO_BENCH_PAIR=0: an image per chroma plane, the r03 form)
        pair = os.environ.get("SCHRO_BENCH_PAIR", "1") != "0"
        (ch, cw) = dims[1]
        if wl.prec == 0:
            # full pel (r06): the references are the plain planes themselves -- no half-pel images, no upsample stage
            # (schrodecoder.c:1596-1601)
            self.hp = [[list(wl.ref[g][r]) for r in range(2)] for g in range(wl.groups)]
            self.up_luma, self.up_chroma = [], []
        elif pair:
            self.hp = [[[ctx.hp_plane(*dims[0])] + [ctx.hp_plane(ch, cw, pair=True)] * 2 for _ in range(2)] for _ in range(wl.groups)]
            self.up_luma = [(wl.ref[g][r][0], self.hp[g][r][0]) for g in range(wl.groups) for r in range(2)]
            self.up_chroma = [((wl.ref[g][r][1], wl.ref[g][r][2]), self.hp[g][r][1]) for g in range(wl.groups) for r in range(2)]
        else:
            self.hp = [[[ctx.hp_plane(h, w) for (h, w) in dims] for _ in range(2)] for _ in range(wl.groups)]
            self.up_luma = [(wl.ref[g][r][0], self.hp[g][r][0]) for g in range(wl.groups) for r in range(2)]
            self.up_chroma = [(wl.ref[g][r][k], self.hp[g][r][k]) for g in range(wl.groups) for r in range(2) for k in (1, 2)]
        self.up_pairs = self.up_luma + self.up_chroma
        self.iwt_pairs, self.obmc_jobs = [], []
        # r04, the combine form (default; SCHRO_BENCH_COMBINE=0: the r03 stage order): the OBMC launches write
        # their PREDICTION (pred planes), the inverse wavelet's last step adds it and writes the picture -- the
        # residual picture is never written or read
        self.combine = wl.combine
        self.iwt_combine, self.pred_jobs, self.iwt_coarse, self.iwt_ll = [], [], [], []
        self.coeff_np, self.mv_np, self.out, self.mv_dev = [], [], [], []
        nmv = 20 * wl.P["x_num_blocks"] * wl.P["y_num_blocks"]
        # (coefficient planes are padded to the transform's multiple of 2^depth: 1080p chroma 540 -> 544 rows)
        pad = lambda n: -(-n // (1 << DEPTH)) * (1 << DEPTH)
        cdims = [(pad(h), pad(w)) for (h, w) in dims]
        self.co_arena = sa.Arena(ctx, sa.Arena.size_of([(d, np.int16) for d in cdims] * wl.frames))
        self.out_arena = sa.Arena(ctx, sa.Arena.size_of([(d, np.uint8) for d in dims] * wl.frames))
        self.mv_arena = sa.Arena(ctx, sa.Arena.size_of([((1, nmv), np.uint8)] * wl.frames))
        base = {}
        for f in range(wl.frames):
            # (vectors uniform in +-16 pel, independently per block -- 64 quarter pels)
            mv = synth.motion_field(wl.P["x_num_blocks"], wl.P["y_num_blocks"], 16 << wl.prec, seed=seed + 2 + f)
            d_mv = self.mv_arena.plane(1, nmv, np.uint8).upload(np.ascontiguousarray(mv).view(np.uint8).reshape(1, -1))
            self.mv_np.append(mv)
            self.mv_dev.append(d_mv)
            co_f, out_f = [], []
            for k, (h, w) in enumerate(dims):
                key = (k, f % 4)        # 4 distinct coefficient sets, uploaded to distinct buffers
                ih, iw = cdims[k]
                if key not in base:
                    base[key] = coeff_plane(ih, iw, seed + 7 * f + k)
                co = base[key]
                d_co = self.co_arena.plane(ih, iw, np.int16).upload(co)
                d_res = ctx.plane(ih, iw, np.int16)
                out = self.out_arena.plane(h, w, np.uint8)
                self.iwt_pairs.append((d_co, d_res))
                if self.combine:
                    d_pred = ctx.plane(h, w, np.uint8)
                    self.iwt_combine.append((d_co, out, d_pred))
                    # the transform in two calls (SchroHipIwtPlane.ll): the levels above 0 into an LL plane ...
                    d_ll = ctx.plane(ih // 2, iw // 2, np.int16)
                    self.iwt_coarse.append((d_co.level_view(1), d_ll))
                    self.iwt_ll.append(d_ll)
                g = min(f // REF_GROUP, wl.groups - 1)
                # (SCHRO_BENCH_ONE_REF=1, a footprint experiment: both references read the same planes)
                r1 = 0 if os.environ.get("SCHRO_BENCH_ONE_REF") == "1" else 1
                # (SCHRO_BENCH_NO_RESIDUAL=1, a pricing experiment: OBMC without its residual -- what fusing the finest
                # wavelet level into its finish could save at most on this side; the pictures are then NOT the workload's)
                no_res = os.environ.get("SCHRO_BENCH_NO_RESIDUAL") == "1"
                self.obmc_jobs.append(sa.obmc_plane(d_mv, wl.P, k, self.hp[g][0][k], self.hp[g][r1][k], None if no_res else d_res, out))
                if self.combine:
                    self.pred_jobs.append(sa.obmc_plane(d_mv, wl.P, k, self.hp[g][0][k], self.hp[g][r1][k], None, d_pred,
                                                        prediction_only=True))
                co_f.append(co)
                out_f.append(out)
            self.coeff_np.append(co_f)
            self.out.append(out_f)


class Workload:
    """`queues` picture batches in flight.  With two queues batch k's OBMC (issue-bound) runs on
    queue 1 beside batch k + 1's upsample + inverse wavelet (HBM-bound) on queue 0; the marks
    are the decoder's stage dependencies (OBMC after its wavelet; a batch's frames are rewritten
    only after the OBMC that read them)."""

    def __init__(self, ctx, frames, seed, queues=2, w=W, h=H, xblen=XBLEN, xbsep=XBSEP, prec=PREC, weights=(1, 1, 1)):
        """w .. prec (r06): the headline's configuration by default; the other pictures a decoder meets -- 1080p, the
        reference encoder's default full-pel vectors, eighth pel, the 24 / 16 block set -- run the same step."""
        self.ctx, self.frames, self.queues = ctx, frames, queues
        self.combine = os.environ.get("SCHRO_BENCH_COMBINE", "1") != "0"
        self.w, self.h, self.prec = w, h, prec
        self.P = synth.motion_params(w, h, xblen, xbsep, prec, tuple(weights), (1, 1))
        dims = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
        self.dims = dims
        # two references (planar u8) per group of REF_GROUP pictures, shared by the batches as between two
        # anchors of a GOP: the work per picture does not depend on the batch size (2 reference
        # upsamples per 8 pictures)
        self.groups = max(1, frames // REF_GROUP)
        self.ref_np_all = [[[synth.picture_u8(h, w, seed=seed + 100 + 1000 * g + 10 * r + k) for k, (h, w) in
                             enumerate(dims)] for r in range(2)] for g in range(self.groups)]
        self.ref_np = self.ref_np_all[0]        # (the first group's: picture 0's references, cpu_baseline's check)
        self.ref = [[[ctx.upload(p) for p in comps] for comps in grp] for grp in self.ref_np_all]
        self.sets = [BatchSet(self, seed + 50 * s) for s in range(queues)]
        s0 = self.sets[0]
        self.coeff_np, self.mv_np, self.out = s0.coeff_np, s0.mv_np, s0.out
        self.k = 0
        self.prev_alone = False

    def step(self, alone=False):
        """alone: this step's launches share the device with no other step's (the steps whose
        launches bench.py brackets with events, so that a kernel's duration is its own).
        Order inside a batch (r03): the inverse wavelet first, then each reference plane set is
        upsampled RIGHT BEFORE the OBMC launch that gathers from it -- luma planes, luma OBMC, chroma
        planes, chroma OBMC -- so the half-pel planes (205 MB per batch, written once, gathered from at
        random) are still in the 256 MB Infinity Cache when they are read, instead of having been
        pushed out by the wavelet's 460 MB of streaming in between (SCHRO_BENCH_ORDER=0: the r02 order)."""
        c, k = self.ctx, self.k
        self.k += 1
        order = int(os.environ.get("SCHRO_BENCH_ORDER", "2"))
        if self.combine:
            # a batch's stages in order on ONE queue, the batches in flight on different queues: the reference planes
            # are upsampled right before the OBMC launch that gathers from them; the transform comes last and writes
            # the pictures
            s = k % self.queues
            b = self.sets[s]
            if self.queues == 2 and os.environ.get("SCHRO_BENCH_PIPE", "whole") == "stages":
                # (A/B form: prediction on queue 1, the transform that adds it on queue 0 -- step k + 1's OBMC beside step k's
                # wavelet; measured no faster than whole batches per queue)
                c.select_queue(1)
                if alone or self.prev_alone:
                    c.queue_wait(1, 0)
                self.prev_alone = alone
                c.queue_wait_mark(8 + s)        # the transform that last read this set's prediction planes
                c.upsample_batch(b.up_luma)
                c.obmc_batch([j for n, j in enumerate(b.pred_jobs) if n % 3 == 0])
                c.upsample_batch(b.up_chroma)
                c.obmc_batch([j for n, j in enumerate(b.pred_jobs) if n % 3])
                c.queue_mark(s)
                c.select_queue(0)
                c.queue_wait_mark(s)
                c.iiwt_batch(b.iwt_combine, DEPTH, FILTER)
                c.queue_mark(8 + s)
                return
            if self.queues > 1:
                c.select_queue(s)
                if alone or self.prev_alone:
                    for other in range(self.queues):
                        if other != s:
                            c.queue_wait(s, other)
                self.prev_alone = alone
            # (A/B, SCHRO_BENCH_SPLIT_IWT=1: measured slower, 0.361 against 0.350 ms per step)
            split = self.queues > 1 and not alone and os.environ.get("SCHRO_BENCH_SPLIT_IWT", "0") == "1"
            if split:
                # ... which do not depend on the prediction: on a queue of their own (2 + s) beside this batch's upsample
                # and OBMC launches -- two launches of a few thousand waves that are latency, not bandwidth
                c.select_queue(2 + s)
                c.queue_wait_mark(12 + s)       # the finest level that last read this set's LL planes
                c.iiwt_batch(b.iwt_coarse, DEPTH - 1, FILTER)
                c.queue_mark(8 + s)
                c.select_queue(s)
            # (A/B, SCHRO_BENCH_SPLIT_OBMC=1, r06: the chroma planes' upsample + OBMC on a queue of their own beside the luma
            # launches -- does a longer run of OBMC workgroups, luma and chroma at once, fill the launches' tails?)
            split_obmc = self.queues > 1 and not alone and os.environ.get("SCHRO_BENCH_SPLIT_OBMC", "0") == "1"
            if split_obmc:
                c.select_queue(2 + s)
                c.queue_wait_mark(12 + s)       # the transform that last read this set's chroma predictions
                if b.up_chroma:
                    c.upsample_batch(b.up_chroma)
                c.obmc_batch([j for n, j in enumerate(b.pred_jobs) if n % 3])
                c.queue_mark(8 + s)
                c.select_queue(s)
            if b.up_luma:
                c.upsample_batch(b.up_luma)
            c.obmc_batch([j for n, j in enumerate(b.pred_jobs) if n % 3 == 0])
            if split_obmc:
                c.queue_wait_mark(8 + s)
                c.iiwt_batch(b.iwt_combine, DEPTH, FILTER)
                c.queue_mark(12 + s)
                if self.queues > 1:
                    c.select_queue(0)
                return
            if b.up_chroma:
                c.upsample_batch(b.up_chroma)
            c.obmc_batch([j for n, j in enumerate(b.pred_jobs) if n % 3])
            if split:
                c.queue_wait_mark(8 + s)
                c.iiwt_batch(b.iwt_combine, 1, FILTER, ll=b.iwt_ll)
                c.queue_mark(12 + s)
            else:
                c.iiwt_batch(b.iwt_combine, DEPTH, FILTER)
            if self.queues > 1:
                c.select_queue(0)
            return

        def obmc_side(b):
            if order == 0:
                c.obmc_batch(b.obmc_jobs)
            elif order == 1:
                c.upsample_batch(b.up_pairs)
                c.obmc_batch(b.obmc_jobs)
            else:
                c.upsample_batch(b.up_luma)
                c.obmc_batch([j for n, j in enumerate(b.obmc_jobs) if n % 3 == 0])
                c.upsample_batch(b.up_chroma)
                c.obmc_batch([j for n, j in enumerate(b.obmc_jobs) if n % 3])
        if self.queues == 1:
            b = self.sets[0]
            if order == 0:
                c.upsample_batch(b.up_pairs)
            c.iiwt_batch(b.iwt_pairs, DEPTH, FILTER)
            obmc_side(b)
            return
        s = k % self.queues
        b = self.sets[s]
        if os.environ.get("SCHRO_BENCH_PIPE", "whole") == "whole":
            # each batch runs all its stages on ONE queue, the two batches in flight on different queues:
            # no cross-queue dependency at all in the steady state
            c.select_queue(s)
            if alone or self.prev_alone:
                for other in range(self.queues):
                    if other != s:
                        c.queue_wait(s, other)
            self.prev_alone = alone
            if order == 0:
                c.upsample_batch(b.up_pairs)
            c.iiwt_batch(b.iwt_pairs, DEPTH, FILTER)
            obmc_side(b)
            c.select_queue(0)
            return
        c.select_queue(0)
        if alone or self.prev_alone:
            c.queue_wait(0, 1)              # the previous batch's OBMC has finished
        self.prev_alone = alone
        c.queue_wait_mark(8 + s)            # the OBMC that last read this batch's frames
        if order == 0:
            c.upsample_batch(b.up_pairs)
        c.iiwt_batch(b.iwt_pairs, DEPTH, FILTER)
        c.queue_mark(s)
        c.select_queue(1)
        c.queue_wait_mark(s)
        obmc_side(b)
        c.queue_mark(8 + s)
        c.select_queue(0)


def batch_kernels(c, b):
    """The kernels of one batch, in the order the workload runs them (the PCIe-inclusive legs)."""
    if b.combine:
        c.upsample_batch(b.up_pairs)
        c.obmc_batch(b.pred_jobs)
        c.iiwt_batch(b.iwt_combine, DEPTH, FILTER)
    else:
        c.upsample_batch(b.up_pairs)
        c.iiwt_batch(b.iwt_pairs, DEPTH, FILTER)
        c.obmc_batch(b.obmc_jobs)


def cpu_baseline(wl, cores, reps=10, check=True):
    """The oracle on `cores` threads, `reps` pictures each (one picture per thread at a time:
    the reference's own picture-level parallelism).  r05: the threads walk through EVERY picture the timed
    steps kept in flight -- the `queues` batch sets x `frames` pictures, 16 by default -- and the first
    result for each is kept: `check` compares all of them with what the device holds (every
    timed output against the CPU restatement).  Returns (Mpix/s, pictures checked, pictures equal, one-thread Mpix/s)."""
    import oracle_lib as O      # cpu_baseline leg only
    O.lib()
    ups = [[[O.UpComp(p, upsample=False) for p in comps] for comps in grp] for grp in wl.ref_np_all]
    pics = [(s, f) for s in range(len(wl.sets)) for f in range(wl.frames)]
    results = {}
    lock = threading.Lock()

    def one(i, n=1):
        for rep in range(n):
            s, f = pics[(i + rep * cores) % len(pics)]
            b = wl.sets[s]
            g = min(f // REF_GROUP, wl.groups - 1)
            outs = []
            for k, (h, w) in enumerate(wl.dims):
                res = O.inverse_iwt(b.coeff_np[f][k], DEPTH, FILTER)
                outs.append(O.motion_render(b.mv_np[f], O.MotionParams(**wl.P), k, ups[g][0][k], ups[g][1][k],
                                            res, w, h))
            with lock:
                if (s, f) not in results:
                    results[(s, f)] = outs

    t0 = time.perf_counter()
    for grp in ups:                        # reference upsampling, once per reference
        for comps in grp:
            ths = [threading.Thread(target=lambda u=u: O.lib().oracle_upcomp_upsample(u.c)) for u in comps]
            [t.start() for t in ths]
            [t.join() for t in ths]
    ths = [threading.Thread(target=one, args=(i, reps)) for i in range(cores)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    checked = equal = 0
    if check:
        for (s, f), outs in sorted(results.items()):
            checked += 1
            equal += all(np.array_equal(wl.sets[s].out[f][k].download(), outs[k]) for k in range(3))
    # one picture on one thread (upsampled references already there): the single-thread figure
    t1 = time.perf_counter()
    one(0)
    single = W * H / (time.perf_counter() - t1) / 1e6
    return cores * reps * W * H / dt / 1e6, checked, equal, single


def coherent_motion_field(nbx, nby, seed):
    """A smooth field -- a pan plus a slow zoom, different for the two references -- with -2 .. +1 quarter-pels of
    noise per block (every quarter-pel phase equally often: the headline's 2.25 taps per window, so that the figure
    differs from the headline by the windows' locality alone) and the headline's mode mix: what a real encoder's
    vectors look like to the gather, next to the headline's independent-per-block worst case."""
    mv = synth.motion_field(nbx, nby, 64, seed=seed)
    mode = mv["flags"] & 3
    yy, xx = np.divmod(np.arange(nbx * nby), nbx)
    n = synth.lcg(4 * nbx * nby, 77 + seed).reshape(4, -1) % 4 - 2
    vec = np.stack([5 + xx // 64 + n[0], -7 + xx // 48 + n[1], 3 + yy // 64 + n[2], 9 - yy // 48 + n[3]], 1).astype(np.int16)
    mv["v"] = np.where((mode == 0)[:, None], mv["v"], vec)
    return mv


def coherent_motion(wl, steps=24):
    """The headline's step with coherent vectors instead of independent ones (outside the timed region; the L1-hit
    sensitivity of the OBMC gather on record): the same pictures, references, coefficients and mode mix."""
    import schroedinger_amd as sa
    c = wl.ctx
    c.select_queue(0)
    c.synchronize()
    saved = []
    keep = []
    for si, b in enumerate(wl.sets):
        saved.append((b.pred_jobs, b.obmc_jobs))
        pj = []
        for f in range(wl.frames):
            mv = coherent_motion_field(wl.P["x_num_blocks"], wl.P["y_num_blocks"], 9000 + 50 * si + f)
            d_mv = c.upload_bytes(mv)
            keep.append(d_mv)
            g = min(f // REF_GROUP, wl.groups - 1)
            for k in range(3):
                pj.append(sa.obmc_plane(d_mv, wl.P, k, b.hp[g][0][k], b.hp[g][1][k], None, b.iwt_combine[3 * f + k][2],
                                        prediction_only=True))
        b.pred_jobs = pj
    for _ in range(6):
        wl.step()
    c.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    c.synchronize()
    dt = (time.perf_counter() - t0) / steps
    c.profile_enable(True)
    c.profile_reset()
    for _ in range(3):
        wl.step(alone=True)
    c.synchronize()             # (both kernel queues: profile_read waits for the selected one only)
    prof = c.profile_read()
    c.profile_enable(False)
    for b, (pj, oj) in zip(wl.sets, saved):
        b.pred_jobs, b.obmc_jobs = pj, oj
    # (back to the headline's pictures: the parity check that follows reads them)
    for _ in range(len(wl.sets)):
        wl.step()
    c.synchronize()
    for d in keep:
        d.free()
    return {"ms_per_step": round(dt * 1e3, 4), "Mpix_per_s": round(wl.frames * W * H / dt / 1e6, 1),
            "obmc_ms_per_step": round(prof["obmc"][0] / 3, 4),
            "vectors": "a pan + a slow zoom per reference, -2 .. +1 quarter-pels of noise per block (all phases equally often), the headline's mode mix",
            "note": "secondary figure, outside the timed region: the headline's vectors are independent per block "
                    "(uniform in +-16 pel), the worst case for the gather"}


def check_picture(wl, s=0, f=0):
    """Picture f of batch set s as the device holds it against the oracle (checker only)."""
    import oracle_lib as O
    b = wl.sets[s]
    g = min(f // REF_GROUP, wl.groups - 1)
    for k, (h, w) in enumerate(wl.dims):
        res = O.inverse_iwt(b.coeff_np[f][k], DEPTH, FILTER)
        u = [O.UpComp(wl.ref_np_all[g][r][k], upsample=wl.prec > 0) for r in range(2)]
        want = O.motion_render(b.mv_np[f], O.MotionParams(**wl.P), k, u[0], u[1], res, w, h)
        if not np.array_equal(b.out[f][k].download(), want):
            return False
    return True


def decode_variant(device, frames=8, steps=24, check=True, queues=2, **geom):
    """The headline's step on another kind of picture (outside the timed region, a context of its own): the whole pixel
    path -- upsample where the precision has one, OBMC prediction, 3-level DD(9,7) transform with the add -- two batches in
    flight, the classes' times from per-launch events of three steps that run alone, the OBMC class against ITS
    algorithmic bytes (1 B written + 1 B per reference used per sample + 20 B per block), one picture against the oracle."""
    import schroedinger_amd as sa
    c = sa.Context(device)
    try:
        wl = Workload(c, frames, seed=4242, queues=queues, **geom)
        for _ in range(8):
            wl.step()
        c.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            wl.step()
        c.synchronize()
        dt = (time.perf_counter() - t0) / steps
        c.profile_enable(True)
        c.profile_reset()
        for _ in range(3):
            wl.step(alone=True)
        c.synchronize()
        prof = c.profile_read()
        c.profile_enable(False)
        w, h = wl.w, wl.h
        samples = frames * (w * h * 3 // 2)
        modes = np.concatenate([m["flags"] & 3 for m in wl.mv_np])
        refs_per_px = float(((modes == 1) | (modes == 2)).mean() + 2 * (modes == 3).mean())
        obmc_bytes = int((1 + refs_per_px) * samples) + 20 * frames * wl.P["x_num_blocks"] * wl.P["y_num_blocks"]
        obmc_ms = prof["obmc"][0] / 3
        out = {"ms_per_step": round(dt * 1e3, 4), "Mpix_per_s": round(frames * w * h / dt / 1e6, 1),
               "pictures_per_step": frames, "width": w, "height": h,
               "blocks": "%dx%d / %dx%d" % (wl.P["xblen_luma"], wl.P["yblen_luma"], wl.P["xbsep_luma"], wl.P["ybsep_luma"]),
               "mv_precision": wl.prec,
               "picture_weights": "%d, %d / 2^%d" % (wl.P["picture_weight_1"], wl.P["picture_weight_2"], wl.P["picture_weight_bits"]),
               "obmc_ms_per_step": round(obmc_ms, 4),
               "obmc_alg_bytes_per_step": obmc_bytes,
               "obmc_frac_of_8TBs": round(obmc_bytes / (obmc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "classes_ms_per_step": {k: round(prof[k][0] / 3, 4) for k in ("obmc", "iiwt_finest", "iiwt_coarse", "upsample")
                                       if prof[k][1]}}
        if check:
            wl.step()               # (the steps above left set 1's pictures last: both sets hold their outputs anyway)
            c.synchronize()
            out["parity"] = "bit-exact vs oracle (picture 0)" if check_picture(wl) else "MISMATCH vs oracle (picture 0)"
        return out
    finally:
        c.close()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def iiwt_1080p(ctx, frames=8, steps=30):
    """BASELINE config 2: 3-level DD(9,7) inverse wavelet of 1920x1080 s16 pictures (4:2:0; the
    chroma coefficient frames are 960x544, padded to a multiple of 2^depth as the reference
    does, schroparams.c:75-92).  Median of `steps` batches of `frames` pictures, HIP events."""
    import schroedinger_amd as sa
    dims = [(1080, 1920), (544, 960), (544, 960)]
    pairs = []
    # the planes of all pictures in two blocks (coefficients, residuals), as a decoder's frame pool is
    # (planes allocated one by one measure the same: SCHRO_BENCH_1080P_ARENA=0)
    arena = os.environ.get("SCHRO_BENCH_1080P_ARENA", "1") != "0"
    a_co = sa.Arena(ctx, sa.Arena.size_of([(d, np.int16) for d in dims] * frames)) if arena else None
    a_res = sa.Arena(ctx, sa.Arena.size_of([(d, np.int16) for d in dims] * frames)) if arena else None
    for f in range(frames):
        for k, (h, w) in enumerate(dims):
            co = coeff_plane(h, w, 300 + 3 * f + k)
            if arena:
                pairs.append((a_co.plane(h, w, np.int16).upload(co), a_res.plane(h, w, np.int16)))
            else:
                pairs.append((ctx.upload(co), ctx.plane(h, w, np.int16)))
    for _ in range(5):
        ctx.iiwt_batch(pairs, DEPTH, FILTER)
    # groups of 8 launch sets between one event pair: the host runs ahead of the device, as in a decoder
    # whose queue is never empty (one launch set per event pair also times the host's table building:
    # 0.051 against 0.039 ms for 8 pictures)
    ts = []
    for _ in range(max(steps // 8, 3)):
        ctx.timer_begin()
        for _ in range(8):
            ctx.iiwt_batch(pairs, DEPTH, FILTER)
        ts.append(ctx.timer_end() / 8)
    # two batches in flight, alternating between the kernel queues: the coarse levels of one (two launches
    # of latency, 8 us each) run beside the finest level of the other
    pairs2 = [(ctx.upload(a.download()), ctx.plane(b.height, b.width, np.int16)) for a, b in pairs]
    for k in range(4):
        ctx.select_queue(k % 2)
        ctx.iiwt_batch(pairs2 if k % 2 else pairs, DEPTH, FILTER)
    ctx.select_queue(0)
    ctx.synchronize()
    t0 = time.perf_counter()
    for k in range(32):
        ctx.select_queue(k % 2)
        ctx.iiwt_batch(pairs2 if k % 2 else pairs, DEPTH, FILTER)
    ctx.select_queue(0)
    ctx.synchronize()
    ms2 = (time.perf_counter() - t0) * 1e3 / 32
    for a, b in pairs + pairs2:
        a.free()
        b.free()
    if arena:
        a_co.block.free()
        a_res.block.free()
    ms = float(np.median(ts))
    samples = frames * sum(h * w for h, w in dims)
    return {"workload": "3-level DD(9,7) IIWT, %d x 1920x1080 4:2:0 s16 per launch set" % frames,
            "median_ms": round(ms, 4), "Mpix_per_s": round(frames * 1920 * 1080 / ms / 1e3, 1),
            "alg_GBs": round(4 * samples / (ms * 1e-3) / 1e9, 1),
            "frac_of_8TBs": round(4 * samples / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "two_batches_in_flight": {"ms": round(ms2, 4), "frac_of_8TBs": round(4 * samples / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}


def iiwt_2160p(wl, reps=20):
    """north_star's own target on its own configuration: the plain 3-level DD(9,7) inverse wavelet of 8 x 2160p 4:2:0
    s16 pictures into a residual frame (no prediction, no combine epilogue), outside the timed region -- tracked
    beside kernels.iiwt_3_levels, which since r04 carries the add of the prediction in its finest level."""
    c = wl.ctx
    c.select_queue(0)
    c.synchronize()
    b0, b1 = wl.sets[0], wl.sets[-1]
    samples = wl.frames * (W * H * 3 // 2)
    for _ in range(4):
        c.iiwt_batch(b0.iwt_pairs, DEPTH, FILTER)
    c.profile_enable(True)
    c.profile_reset()
    for _ in range(reps):
        c.iiwt_batch(b0.iwt_pairs, DEPTH, FILTER)
    prof = c.profile_read()
    c.profile_enable(False)
    fin, coarse = prof["iiwt_finest"][0] / reps, prof["iiwt_coarse"][0] / reps
    c.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        c.iiwt_batch(b0.iwt_pairs, DEPTH, FILTER)
    c.synchronize()
    wall1 = (time.perf_counter() - t0) * 1e3 / reps
    for k in range(4):
        c.select_queue(k % 2)
        c.iiwt_batch((b1 if k % 2 else b0).iwt_pairs, DEPTH, FILTER)
    c.select_queue(0)
    c.synchronize()
    t0 = time.perf_counter()
    for k in range(2 * reps):
        c.select_queue(k % 2)
        c.iiwt_batch((b1 if k % 2 else b0).iwt_pairs, DEPTH, FILTER)
    c.select_queue(0)
    c.synchronize()
    wall2 = (time.perf_counter() - t0) * 1e3 / (2 * reps)
    ms = fin + coarse

    def frac(t, bytes_per_sample=4):
        return round(bytes_per_sample * samples / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return {"workload": "3-level DD(9,7) IIWT, %d x 3840x2160 4:2:0 s16 per launch set, residual frame out" % wl.frames,
            "kernels_ms": round(ms, 4), "finest_ms": round(fin, 4), "coarse_ms": round(coarse, 4),
            "alg_GBs": round(4 * samples / (ms * 1e-3) / 1e9, 1), "frac_of_8TBs": frac(ms),
            "read_frac_of_8TBs": frac(ms, 2), "Mpix_per_s": round(wl.frames * W * H / ms / 1e3, 1),
            "wall_ms_one_batch": round(wall1, 4), "wall_frac_of_8TBs_one_batch": frac(wall1),
            "wall_ms_two_batches_in_flight": round(wall2, 4), "wall_frac_of_8TBs_two_batches": frac(wall2),
            "note": "kernels_ms: the three launches' own durations (per-launch events); 4 B per sample (2 read + 2 written), "
                    "SURVEY 8(d); read_frac: the 2 B per sample read side alone"}


def iiwt_s32_2160p(device, frames=8, reps=12):
    """r06: the 3-level transform of s32 frames -- what pictures of more than 8 bits get in the core
    syntax (schrodecoder.c:350-352), the 10-bit professional case -- for DD(9,7) and LeGall(5,3) on 8 x 2160p 4:2:0:
    8 B per sample (4 read + 4 written).  A context of its own, outside the timed region."""
    import schroedinger_amd as sa
    c = sa.Context(device)
    try:
        dims = [(H, W), (H // 2, W // 2), (H // 2, W // 2)]
        samples = frames * (W * H * 3 // 2)
        base = [coeff_plane(h, w, 31 + k).astype(np.int32) for k, (h, w) in enumerate(dims)]
        pairs = [(c.upload(base[k]), c.plane(h, w, np.int32)) for _ in range(frames) for k, (h, w) in enumerate(dims)]
        out = {}
        for filt, name in ((0, "dd97"), (1, "legall53")):
            for _ in range(3):
                c.iiwt_batch(pairs, DEPTH, filt)
            c.profile_enable(True)
            c.profile_reset()
            for _ in range(reps):
                c.iiwt_batch(pairs, DEPTH, filt)
            c.synchronize()
            prof = c.profile_read()
            c.profile_enable(False)
            ms = (prof["iiwt_finest"][0] + prof["iiwt_coarse"][0]) / reps
            out[name] = {"kernels_ms": round(ms, 4), "finest_ms": round(prof["iiwt_finest"][0] / reps, 4),
                         "alg_GBs": round(8 * samples / (ms * 1e-3) / 1e9, 1),
                         "frac_of_8TBs": round(8 * samples / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "Mpix_per_s": round(frames * W * H / ms / 1e3, 1)}
        out["workload"] = "3-level IIWT, %d x 3840x2160 4:2:0 s32 per launch set, 8 B per sample (4 read + 4 written)" % frames
        return out
    finally:
        c.close()


def quantised_handover(h, w, depth, stride, seed):
    """A synthetic core-syntax hand-over for one s16 coefficient plane: every sub-band cut into
    up to 8x8 codeblocks (schrodecoder.c:3572-3596 geometry), a level-dependent share of them
    zero codeblocks, Laplacian quantised values one byte each in the others.  Returns the values
    blob and the SchroHipCodeblock tuples for a device plane of row pitch `stride`."""
    rng = np.random.default_rng(seed)
    blob, cbs = [], []
    nbytes = 0
    for index in range(1 + 3 * depth):
        # schro_subband_get_position (schroparams.c:355-367), schro_subband_get_frame_data (:319-352)
        position = 0 if index == 0 else (((index - 1) // 3) << 2) | ((index - 1) % 3 + 1)
        level = position >> 2                      # 0: LL and the coarsest detail bands ... depth - 1: finest
        shift = depth - level
        bh, bw = h >> shift, w >> shift
        row0 = ((1 << shift) >> 1) if position & 2 else 0
        col0 = bw if position & 1 else 0
        ncx, ncy = min(8, bw), min(8, bh)
        p_zero = 0.0 if index == 0 else min(0.75, 0.25 * level + 0.25)
        scale = 24.0 / (1 << (2 * level)) if index else 40.0
        for cy in range(ncy):
            y0, y1 = (bh * cy) // ncy, (bh * (cy + 1)) // ncy
            for cx in range(ncx):
                x0, x1 = (bw * cx) // ncx, (bw * (cx + 1)) // ncx
                dst_off = (row0 + (y0 << shift)) * stride + (col0 + x0) * 2
                if rng.random() < p_zero:
                    cbs.append((dst_off, stride << shift, x1 - x0, y1 - y0, -1, 0, 20))
                    continue
                q = np.clip(np.rint(rng.laplace(0.0, scale, (y1 - y0, x1 - x0))), -127, 127).astype(np.int8)
                cbs.append((dst_off, stride << shift, x1 - x0, y1 - y0, nbytes, 1, 20))
                blob.append(q.reshape(-1))
                nbytes += q.size
    return np.concatenate(blob).view(np.uint8), cbs


class HostSide:
    """The host's side of one picture batch for the PCIe-inclusive figures, in PINNED host memory
    (schro_hip_host_alloc: what a decoder whose frames come from schro_memory_domain_new_hip_host ()
    hands over): ONE block each for the coefficient frames (dense hand-over) or the quantised values
    of the non-zero codeblocks (quantised hand-over; their C tables built once), the motion vectors
    and the output pictures -- mirrors of the batch's device arenas, one copy per kind and step."""

    def __init__(self, wl, b, quantised, seed):
        c = wl.ctx
        self.mv = c.host_array((1, b.mv_arena.nbytes), np.uint8)
        self.mv[...] = b.mv_arena.block.download()
        self.out = c.host_array((1, b.out_arena.nbytes), np.uint8)
        self.h2d, self.d2h = b.mv_arena.used, b.out_arena.used
        self.co = self.blob = self.d_blob = None
        self.hand = []
        if not quantised:
            self.co = c.host_array((1, b.co_arena.nbytes), np.uint8)
            self.co[...] = b.co_arena.block.download()
            self.h2d += b.co_arena.used
            return
        import schroedinger_amd as sa
        parts, off = [], 0
        for f in range(wl.frames):
            for k, (h, w) in enumerate(wl.dims):
                dst = b.iwt_pairs[3 * f + k][0]
                blob, cbs = quantised_handover(h, w, DEPTH, dst.stride, seed + 3 * f + k)
                parts.append((off, blob, dst, cbs))
                off += (blob.size + 255) // 256 * 256
        self.blob = c.host_array((1, off), np.uint8)
        self.d_blob = sa.Arena(c, off)
        for o, blob, dst, cbs in parts:
            self.blob[0, o:o + blob.size] = blob
            dev = self.d_blob.plane(1, blob.size, np.uint8)
            assert dev.ptr == self.d_blob.ptr + o
            # the C table is built once per picture geometry, not per step
            self.hand.append((dst, dev, c.codeblock_table(cbs)))
            self.h2d += blob.size + 24 * len(cbs)
        # r04: the geometry of the codeblock records lives on the device (one plan for the workload); a step hands
        # over this batch's plane array -- pointers + the records as they are -- and the values
        jobs = [(dst, dev, tab, False) for dst, dev, tab in self.hand]
        if not hasattr(wl, "dq_plan"):
            wl.dq_plan = c.dequant_plan(jobs, 0)
        self.dq_planes = wl.dq_plan.planes(jobs)

    def view(self, b, f, k):
        """Output plane (f, k) of the batch as it came down."""
        p = b.out[f][k]
        o = p.ptr - b.out_arena.ptr
        return self.out[0, o:o + p.nbytes].reshape(p.height, p.stride)[:, :p.width]


def pcie_pipeline(wl, quantised, steps=12, warmup=4, form=None):
    """The step with the host hand-over in it, as a three-stage pipeline on the context's four queues
    (include/schro_hip.h, asynchronous transfers): batch k + 1's coefficients (dense s16 frames, or
    quantised values for schro_hip_dequant_batch) and vectors go up on the H2D queue while batch k's
    kernels run on a kernel queue and batch k - 1's pictures come down on the D2H queue; marks carry
    the dependencies, the host thread never waits inside the loop.  Three batches' buffers; a batch's
    planes of one kind are one block on either side: one copy per kind and step."""
    c = wl.ctx
    c.select_queue(0)
    c.synchronize()
    # three batches' buffers: one going up, one in the kernels, one coming down (with two the download of
    # batch k holds back the kernels of batch k + 2 and the copy engines idle a quarter of the time)
    form = form or os.environ.get("SCHRO_BENCH_PCIE", "handover")
    per_queue, handover = form == "queues", form == "handover"
    nq = int(os.environ.get("SCHRO_BENCH_PCIE_QUEUES", "3"))
    want = max(3, nq) if per_queue else 3
    if not hasattr(wl, "pcie_sets"):
        wl.pcie_sets = list(wl.sets)
    wl.pcie_sets += [BatchSet(wl, 4242 + 50 * n) for n in range(len(wl.pcie_sets), want)]
    sets = wl.pcie_sets[:want]
    nb = len(sets)
    hs = [HostSide(wl, b, quantised, 900 + 100 * i) for i, b in enumerate(sets)]

    # r04, second form ("queues"): a batch's copies and kernels in order on ONE queue (batch k on queue k % 3), nothing crosses queues.
    # On this runtime a hipMemcpyAsync whose queue waits for another queue's event can block the calling
    # thread until that event (quantised hand-over: 1.4 of the step's 2.3 ms spent inside the two copy calls,
    # scripts/pcie_host_time.py; the dense hand-over does not show it) -- SCHRO_BENCH_PCIE=copyq is the r03 form
    # on the copy queues with marks.
    def step_one_queue(k):
        i = k % nb
        b, h = sets[i], hs[i]
        c.select_queue(i % nq)
        if quantised:
            h.d_blob.block.upload_async(h.blob)
        else:
            b.co_arena.block.upload_async(h.co)
        b.mv_arena.block.upload_async(h.mv)
        if quantised:
            wl.dq_plan.run(planes=h.dq_planes)
        batch_kernels(c, b)
        b.out_arena.block.download_async(h.out)

    # r04, third form ("handover"): the copy queues of the r03 form, but the host hands over picture batch k - 2 -- waits for
    # ITS download, the one wait a decoder has anyway -- before it enqueues anything of step k.  Every event a copy of step k
    # then depends on has already fired when the copy is enqueued (the set's last readers are the kernels of step k - 3, the
    # download of step k - 1 follows kernels that ran while download k - 2 was on the bus), so no copy call blocks, and the copy
    # engines overlap as in the r03 form.
    waited = [0.0]

    def step_handover(k):
        i = k % nb
        b, h = sets[i], hs[i]
        if k >= 2:
            tw = time.perf_counter()
            c.queue_mark_synchronize(12 + (k - 2) % nb)
            waited[0] += time.perf_counter() - tw
        c.select_queue(c.QUEUE_H2D)
        if quantised:
            h.d_blob.block.upload_async(h.blob)
        else:
            b.co_arena.block.upload_async(h.co)
        b.mv_arena.block.upload_async(h.mv)
        c.queue_mark(i)
        c.select_queue(k % 2)
        c.queue_wait_mark(i)
        if quantised:
            wl.dq_plan.run(planes=h.dq_planes)
        batch_kernels(c, b)
        c.queue_mark(4 + i)
        if k >= 1:
            j = (k - 1) % nb
            c.select_queue(c.QUEUE_D2H)
            c.queue_wait_mark(4 + j)
            sets[j].out_arena.block.download_async(hs[j].out)
            c.queue_mark(12 + j)

    def step(k):
        if handover:
            return step_handover(k)
        if per_queue:
            return step_one_queue(k)
        i = k % nb
        b, h = sets[i], hs[i]
        c.select_queue(c.QUEUE_H2D)
        c.queue_wait_mark(8 + i)                # the kernels that last read this batch's inputs
        if quantised:
            h.d_blob.block.upload_async(h.blob)
        else:
            b.co_arena.block.upload_async(h.co)
        b.mv_arena.block.upload_async(h.mv)
        c.queue_mark(i)
        c.select_queue(k % 2)
        c.queue_wait_mark(i)
        c.queue_wait_mark(12 + i)               # the download that last read this batch's pictures
        if quantised:
            if os.environ.get("SCHRO_BENCH_DEQUANT_PLAN", "1") != "0":
                wl.dq_plan.run(planes=h.dq_planes)
            else:               # (the r03 form: every codeblock record turned into a job on the host, every step)
                c.dequant_batch([(dst, dev, tab, False) for dst, dev, tab in h.hand], 0)
        batch_kernels(c, b)
        c.queue_mark(8 + i)
        c.queue_mark(4 + i)
        c.select_queue(c.QUEUE_D2H)
        c.queue_wait_mark(4 + i)
        b.out_arena.block.download_async(h.out)
        c.queue_mark(12 + i)

    for k in range(warmup):
        step(k)
    c.select_queue(0)
    c.synchronize()
    waited[0] = 0.0
    t0 = time.perf_counter()
    for k in range(warmup, warmup + steps):
        step(k)
    t_host = time.perf_counter() - t0
    if handover:        # the last step's pictures
        c.select_queue(c.QUEUE_D2H)
        j = (warmup + steps - 1) % nb
        c.queue_wait_mark(4 + j)
        sets[j].out_arena.block.download_async(hs[j].out)
    c.select_queue(0)
    c.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # the pictures that came down are the ones the device holds
    ok = all(np.array_equal(hs[0].view(sets[0], 0, kk), sets[0].out[0][kk].download()) for kk in range(3))
    res = {"ms_per_step": round(dt * 1e3, 3), "Mpix_per_s": round(wl.frames * W * H / dt / 1e6, 1),
           "h2d_MB": round(hs[0].h2d / 1e6, 1), "d2h_MB": round(hs[0].d2h / 1e6, 1),
           "host_GBs": round((hs[0].h2d + hs[0].d2h) / dt / 1e9, 1), "host_enqueue_ms_per_step": round(t_host / steps * 1e3, 3),
           "downloaded_equals_device": bool(ok),
           "queues": "a batch's copies and kernels in order on one queue, %d batches on %d queues" % (nb, nq) if per_queue
                     else "copies on the H2D / D2H queues, marks; the host hands over batch k - 2 before it enqueues step k" if handover
                     else "copies on the H2D / D2H queues beside the kernels, marks for the dependencies",
           "note": "pinned host buffers, asynchronous copies, one copy per kind and step; %d steps in steady state; "
                   "never `value`" % steps}
    if handover:
        res["host_wait_for_handover_ms_per_step"] = round(waited[0] / steps * 1e3, 3)
        res["host_enqueue_ms_per_step"] = round((t_host - waited[0]) / steps * 1e3, 3)
    if quantised:
        dense = sum(co.nbytes for cf in sets[0].coeff_np for co in cf)
        res["share_of_dense_coefficient_bytes"] = round((hs[0].h2d - sets[0].mv_arena.used) / dense, 3)
        res["note"] += ("; synthetic quantised hand-over: up to 8x8 codeblocks per sub-band, 35-75 % of the finer "
                        "levels' codeblocks zero, Laplacian values one byte each; C codeblock tables built once; "
                        "overwrites the batches' coefficient frames")
    return res


def lowdelay_8k(ctx, npic=4, steps=8):
    """BASELINE config 5: VC-2 low-delay 10-bit 4:2:2 7680x4320 -- s32 coefficients, slices of 32x8
    luma samples in 155 bytes, 3-level Haar (no shift).  Per picture: slice decode + dequantisation,
    DC prediction of the LL bands, inverse wavelet; batches of `npic` pictures take turns on three
    queues (SCHRO_BENCH_LD_QUEUES; the DC prediction is a dependency chain on a few CUs: it runs beside
    the other batches' slices / wavelet -- one queue 0.220, two 0.197, three 0.187 ms per picture).  Slices come from tests/synth.py's writer (not from oracle/); a sample
    of slices is checked against that writer's values here, the whole path against the oracle in
    tests/test_gpu_lowdelay.py."""
    import schroedinger_amd as sa
    Wl, Hl, depth, filt = 7680, 4320, 3, 3
    P = synth.lowdelay_params(Wl, Hl, (1, 0), depth, 32, 8, 155, 1)
    data, kind, made = synth.lowdelay_picture(P, 3)
    tables = json.load(open(os.path.join(ROOT, "tests", "golden", "quant_tables.json")))
    dims = [(P["iwt_luma_height"], P["iwt_luma_width"])] + [(P["iwt_chroma_height"], P["iwt_chroma_width"])] * 2

    def batch():
        pics = []
        for _ in range(npic):
            pics.append((ctx.upload_bytes(data), [ctx.plane(h, w, np.int32) for (h, w) in dims],
                         [ctx.plane(h, w, np.int32) for (h, w) in dims], ctx.plane(Hl, 16 * (-(-Wl // 6)), np.uint8)))
        return (pics, [(sl, co) for sl, co, _, _ in pics], [(c, p) for _, co, px, _ in pics for c, p in zip(co, px)],
                [(px, 1, 0, v210, Wl, Hl) for _, _, px, v210 in pics], [(co, 1, 0, v210, Wl, Hl) for _, co, _, v210 in pics])
    nq = int(os.environ.get("SCHRO_BENCH_LD_QUEUES", "3"))
    sets = [batch() for _ in range(nq)]
    pics, jobs, pairs, packs, fused = sets[0]
    ctx.select_queue(0)
    ctx.lowdelay_batch(jobs, P)
    ctx.synchronize()
    # a sample of slices against what the writer put in: every sub-band but the (DC-predicted) LL band
    ok, nx, ny = True, P["n_horiz_slices"], P["n_vert_slices"]
    for (sy, sx) in ((0, 0), (ny // 2, nx // 3), (ny - 1, nx - 1)):
        base, vals = made[int(kind[sy, sx])]
        for comp, (h, w) in enumerate(dims):
            rows = pics[-1][1][comp].download()
            for index in range(1, 1 + 3 * depth):
                _, c0, r0, step, bw, bh = synth.subband_geometry(w, h, depth, index)
                x0, x1, y0, y1 = bw * sx // nx, bw * (sx + 1) // nx, bh * sy // ny, bh * (sy + 1) // ny
                got = rows[r0 + step * y0:r0 + step * y1:step, c0 + x0:c0 + x1]
                ok = ok and np.array_equal(got, synth.lowdelay_expected_band(P, comp, index, base, vals, tables))
    for _ in range(2):
        ctx.lowdelay_batch(jobs, P)
        ctx.iiwt_batch(pairs, depth, filt)
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(steps):
        ctx.lowdelay_batch(jobs, P)
        ctx.iiwt_batch(pairs, depth, filt)
    prof = ctx.profile_read()
    # r05: the copy-out the application receives (v210, 10-bit 4:2:2: schrodecoder.c:2011-2052) -- as its own pass over the
    # pixel frame, and as the epilogue of the transform (schro_hip_iiwt_pack_v210_batch: no pixel frame)
    ctx.profile_reset()
    for _ in range(steps):
        ctx.pack_v210_batch(packs)
    prof_pack = ctx.profile_read()
    two_pass_bytes = pics[0][3].download()[:64].copy()
    ctx.profile_reset()
    for _ in range(steps):
        ctx.iiwt_pack_v210_batch(fused, depth, filt)
    prof_fused = ctx.profile_read()
    ctx.profile_enable(False)
    fused_equal = bool(np.array_equal(pics[0][3].download()[:64], two_pass_bytes))

    def run(step, reps):
        for k in range(4):
            step(k)
        ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(reps):
            step(k)
        ctx.synchronize()
        return (time.perf_counter() - t0) * 1e3 / reps

    def step(k):
        _, jobs_q, pairs_q, _, _ = sets[k % nq]
        ctx.select_queue(k % nq)
        ctx.lowdelay_batch(jobs_q, P)
        ctx.iiwt_batch(pairs_q, depth, filt)

    def step_two_pass(k):
        _, jobs_q, pairs_q, packs_q, _ = sets[k % nq]
        ctx.select_queue(k % nq)
        ctx.lowdelay_batch(jobs_q, P)
        ctx.iiwt_batch(pairs_q, depth, filt)
        ctx.pack_v210_batch(packs_q)

    def step_fused(k):
        _, jobs_q, _, _, fused_q = sets[k % nq]
        ctx.select_queue(k % nq)
        ctx.lowdelay_batch(jobs_q, P)
        ctx.iiwt_pack_v210_batch(fused_q, depth, filt)
    wall = run(step, 2 * steps)
    wall_two = run(step_two_pass, 2 * steps)
    wall_fused = run(step_fused, 2 * steps)
    ctx.select_queue(0)
    for pics_q, _, _, _, _ in sets:
        for sl, co, px, v210 in pics_q:
            sl.free()
            v210.free()
            [p.free() for p in co + px]
    samples = sum(h * w for h, w in dims)
    per = {k: ms / steps / npic for k, (ms, n) in prof.items() if n}
    iiwt = per.get("iiwt_finest", 0) + per.get("iiwt_coarse", 0)
    # algorithmic bytes per picture: compressed slices in, 4 B per coefficient out; wavelet: 8 B per s32 sample
    # (4 read + 4 written) -- the one-pass Haar kernel moves exactly that
    return {"workload": "7680x4320 4:2:2 s32 low-delay, 32x8 slices of 155 bytes, 3-level Haar, %d pictures per launch, "
                        "%d batches taking turns on as many queues" % (npic, nq),
            "slices_frac_of_8TBs": round((4 * samples + data.size) / (per.get("slices", 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "ms_per_picture": round(wall / npic, 4), "Mpix_per_s": round(Wl * Hl * npic / wall / 1e3, 1),
            "kernels_ms_per_picture": {"slices": round(per.get("slices", 0), 4), "dc_predict": round(per.get("dc_predict", 0), 4),
                                       "iiwt_3_levels": round(iiwt, 4)},
            "alg_GBs": {"slices": round((4 * samples + data.size) / (per.get("slices", 1) * 1e-3) / 1e9, 1),
                        "iiwt_3_levels": round(8 * samples / (max(iiwt, 1e-9) * 1e-3) / 1e9, 1)},
            "iiwt_frac_of_8TBs": round(8 * samples / (max(iiwt, 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "compressed_MB_per_picture": round(data.size / 1e6, 1), "coefficient_MB_per_picture": round(4 * samples / 1e6, 1),
            "sample_slices_vs_writer": "bit-exact" if ok else "MISMATCH",
            # r05: with the v210 copy-out (what an application receives).  Fused: slices + DC + (three Haar levels with the copy-out
            # as their epilogue); two passes: + the s32 pixel frame written, read back and packed
            "ms_per_picture_with_copy_out": round(wall_fused / npic, 4),
            "Mpix_per_s_with_copy_out": round(Wl * Hl * npic / wall_fused / 1e3, 1),
            "ms_per_picture_two_passes_with_copy_out": round(wall_two / npic, 4),
            "copy_out_kernels_ms_per_picture": {"pack_v210_from_the_pixel_frame": round(prof_pack.get("convert", (0, 0))[0] / steps / npic, 4),
                                                "haar_3_levels_with_v210_epilogue": round(prof_fused.get("iiwt_finest", (0, 0))[0] / steps / npic, 4)},
            "copy_out_alg_GBs": {"haar_3_levels_with_v210_epilogue":
                                 round((4 * samples + 16 * (-(-Wl // 6)) * Hl) / (max(prof_fused.get("iiwt_finest", (0, 0))[0], 1e-9) / steps / npic * 1e-3) / 1e9, 1)},
            "v210_MB_per_picture": round(16 * (-(-Wl // 6)) * Hl / 1e6, 1),
            "fused_first_bytes_equal_two_passes": fused_equal}


def frame_layer_2160p(npic=24):
    """The boundary the reference would actually bind (VERDICT r03 missing 1): tests/c/stage_loop.c, a compiled C
    caller, decodes 2160p inter pictures through the SchroFrame-shaped stage calls -- x_wavelet_transform ->
    x_upsample (2 references per 8 pictures) -> x_render_motion -> x_combine -- (i) under the reference's
    contract (one picture at a time, every stage call complete on return, host transform frames and vector
    arrays: schroasync-pthread.c:320-328) and (ii) as INTEGRATION.md 3a pipelines them (stage completion off,
    pinned host frames, copy queues, five pictures in flight) and (iii, r05) the same with the quantised hand-over
    (schro_hipframe_dequantise: 17 % of the dense coefficient bytes cross the bus).  All include the host hand-over (25 MB of
    coefficients up, 12 MB of picture down per picture): compare with pcie_inclusive, not with `value`."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "tests", "c", "_build", "stage_loop")
    if not os.path.exists(exe):
        return {"error": "tests/c/_build/stage_loop not built (__graft_entry__.build ())"}
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([exe, d, str(W), str(H), str(npic), "0"], capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        return {"error": "stage_loop rc %d: %s" % (p.returncode, (p.stdout + p.stderr)[-300:])}
    return json.loads(p.stdout.strip().splitlines()[-1])


def extra_kernels(wl):
    """The remaining kernel classes of the path on the headline's pictures, one profiled launch set each:
    core-syntax dequantisation (schro_hip_dequant_batch), intra convert s16 -> u8, packed copy-out (UYVY)."""
    c, b = wl.ctx, wl.sets[0]
    c.select_queue(0)
    c.synchronize()
    hand = []
    for f in range(wl.frames):
        for k, (h, w) in enumerate(wl.dims):
            dst = b.iwt_pairs[3 * f + k][0]
            blob, cbs = quantised_handover(h, w, DEPTH, dst.stride, 700 + 3 * f + k)
            hand.append((dst, c.upload(blob.reshape(1, -1)), c.codeblock_table(cbs), blob.size))
    conv = [(b.iwt_pairs[n][1], b.out[n // 3][n % 3]) for n in range(3 * wl.frames)]
    packed = [c.plane(H, 2 * W, np.uint8) for _ in range(wl.frames)]
    packs = [(b.out[f], 1, 1, packed[f], W, H, 0x101) for f in range(wl.frames)]
    res = {}
    samples = wl.frames * (W * H * 3 // 2)
    # ms_per_step: the launches' own durations (per-launch events, as for the headline's classes);
    # call_ms: from the call to the last kernel's end on the queue -- for the dequantisation that is the
    # host building 15 k codeblock records (Python + the C loop), not the kernel
    dq_jobs = [(d, v, t, False) for d, v, t, _ in hand]
    dq_plan = c.dequant_plan(dq_jobs, 0)
    dq_planes = dq_plan.planes(dq_jobs)
    for name, cls, fn, alg in (("dequant", "dequant", lambda: dq_plan.run(planes=dq_planes),
                                2 * samples + sum(n for _, _, _, n in hand)),
                               ("dequant_batch_call", "dequant", lambda: c.dequant_batch(dq_jobs, 0),
                                2 * samples + sum(n for _, _, _, n in hand)),
                               ("convert", "convert", lambda: c.convert_u8_batch(conv), 3 * samples),
                               ("pack_uyvy", "convert", lambda: c.pack_u8_batch(packs), samples + 2 * wl.frames * W * H)):
        for _ in range(2):
            fn()
        ts = []
        for _ in range(5):
            c.timer_begin()
            fn()
            ts.append(c.timer_end())
        c.profile_enable(True)
        c.profile_reset()
        for _ in range(5):
            fn()
        ms = c.profile_read()[cls][0] / 5
        c.profile_enable(False)
        res[name] = {"ms_per_step": round(ms, 4), "alg_GBs": round(alg / (ms * 1e-3) / 1e9, 1),
                     "call_ms": round(float(np.median(ts)), 4)}
    c.synchronize()
    dq_plan.free()
    for _, v, _, _ in hand:
        v.free()
    [p.free() for p in packed]
    # (the dequantisation overwrote batch 0's coefficient frames and the conversion its pictures: restore)
    for n, (d_co, _) in enumerate(b.iwt_pairs):
        d_co.upload(b.coeff_np[n // 3][n % 3])
    wl.queues_saved = wl.queues
    wl.queues = 1
    wl.step()
    wl.queues = wl.queues_saved
    c.synchronize()
    return res


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (sysfs cpulist syntax)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_cpus(index, sysfs="/sys"):
    """The NUMA node of the index-th AMD GPU (drm cards with vendor 0x1002, in PCI address order -- the order HIP
    enumerates them in) and that node's CPUs, from sysfs alone: called BEFORE the first HIP call.  (None, []) when
    the box does not say (no such card, numa_node -1, no cpulist)."""
    import glob
    import re
    cards = []
    for d in glob.glob(os.path.join(sysfs, "class", "drm", "card*")):
        if not re.fullmatch(r"card\d+", os.path.basename(d)):
            continue
        dev = os.path.join(d, "device")
        try:
            if open(os.path.join(dev, "vendor")).read().strip().lower() != "0x1002":
                continue
            node = int(open(os.path.join(dev, "numa_node")).read().strip())
        except (OSError, ValueError):
            continue
        cards.append((os.path.basename(os.path.realpath(dev)), node))
    cards.sort()
    if index < 0 or index >= len(cards) or cards[index][1] < 0:
        return None, []
    node = cards[index][1]
    try:
        cpus = parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")).read())
    except (OSError, ValueError):
        cpus = []
    return node, cpus


def bind_to_gpu_numa_node(index, sysfs="/sys", setaffinity=None, getaffinity=None):
    """Bind this process (its threads are created later and inherit the mask) to the CPUs of the GPU's NUMA node,
    intersected with what it may use already: the host side of a rank -- pinned buffers, enqueue thread, the
    exec-domain threads -- then sits next to its GPU's PCIe root (the expected scaling limit is the
    host side).  Returns what it did, for the bench line."""
    setaffinity = setaffinity or (lambda cpus: os.sched_setaffinity(0, cpus))
    getaffinity = getaffinity or (lambda: os.sched_getaffinity(0))
    node, cpus = gpu_numa_cpus(index, sysfs)
    if node is None or not cpus:
        return {"numa_node": node, "bound": False}
    try:
        allowed = sorted(set(cpus) & set(getaffinity()))
        if not allowed:
            return {"numa_node": node, "bound": False, "reason": "none of the node's CPUs is in this process's mask"}
        setaffinity(allowed)
    except (OSError, AttributeError) as e:
        return {"numa_node": node, "bound": False, "reason": str(e)}
    return {"numa_node": node, "bound": True, "cpus": len(allowed)}


def spawn_ranks(n, argv, popen=None):
    """Start n rank processes of this script (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, gloo
    rendezvous on 127.0.0.1), wait for all of them, return the largest exit code.  Rank 0's
    stdout is this process's stdout, so the one JSON line comes out as usual."""
    import subprocess
    popen = popen or subprocess.Popen
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                           stdout=None if r == 0 else subprocess.DEVNULL))
    codes = [p.wait() for p in procs]
    if any(codes):
        sys.stderr.write("bench.py: rank exit codes %s\n" % codes)
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs (= rank processes) of this node; default: WORLD_SIZE under a launcher, else 1")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=8, help="pictures per step per GPU")
    ap.add_argument("--queues", type=int, default=2, choices=(1, 2, 3),
                    help="picture batches in flight per GPU (2: OBMC of one beside the wavelet of the next)")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed steps for this long before the warm-up steps (GPU clocks / caches in their steady state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed workload: no CPU baseline, no 1080p / PCIe-inclusive extras "
                         "(profiler runs: every launch in the trace is a launch of the headline step)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="threads for the CPU baseline (0: auto)")
    ap.add_argument("--profile-steps", type=int, default=-1,
                    help="time the launches of the LAST n timed steps with HIP events (default: max (3, steps / 8)); "
                         "such a step runs alone on the device (no other batch beside it), so a kernel's duration "
                         "is its own; as one block at the end the two-queue pipeline drains once, not per sample")
    ap.add_argument("--profile-every", type=int, default=0,
                    help="(r02 form) bracket every n-th timed step instead; n > steps: no samples")
    args = ap.parse_args()
    if args.gpus is None:           # under a launcher the world size is the number of GPUs
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.
        # It has not touched HIP (nothing GPU-side is imported above), starts N fresh rank
        # processes -- one per device, the shape of the reference's one exec-domain thread per
        # device, schroasync-pthread.c:362-390 -- and exits with the worst of their codes.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # r05: with several ranks on a node each rank's host threads go to its GPU's NUMA node -- before anything touches
    # HIP (the runtime's own threads inherit the mask).  One rank keeps the whole mask (SCHRO_BENCH_NUMA=1 binds it too).
    affinity = None
    if (world > 1 or os.environ.get("SCHRO_BENCH_NUMA") == "1") and os.environ.get("SCHRO_BENCH_NUMA") != "0" \
            and os.environ.get("SCHRO_BENCH_SHARE_DEVICE") != "1":
        affinity = bind_to_gpu_numa_node(local_rank)
    import schroedinger_amd as sa
    ndev = sa.device_count()
    if ndev < 1:
        raise SystemExit("bench.py rank %d: needs a HIP device; there is no CPU fallback" % rank)
    share = os.environ.get("SCHRO_BENCH_SHARE_DEVICE") == "1"   # rehearsals of the N > 1 path on a one-GPU box
    if world > ndev and not share:
        raise SystemExit("bench.py rank %d: %d ranks but only %d HIP device(s); one rank per GPU"
                         % (rank, world, ndev))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        import datetime
        # (a rank that died must not leave the others at the rendezvous for gloo's default half hour)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))

    ctx = sa.Context(local_rank % ndev if share else local_rank)
    wl = Workload(ctx, args.frames, seed=1 + 1000 * rank, queues=args.queues)

    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()

    # The card idles at low clocks while the host builds the inputs: a stretch of the same steps brings
    # it to the state a decoder that runs continuously is in, before the W warm-up steps (untimed, like
    # them; --prewarm-ms 0 leaves it out)
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for _ in range(8):
            wl.step()
        ctx.synchronize()
    for _ in range(args.warmup):
        wl.step()
    ctx.profile_enable(True)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    profiled_steps = 0
    n_samp = max(1, min(args.steps, args.profile_steps if args.profile_steps >= 0 else max(3, args.steps // 8)))
    for i in range(args.steps):
        sample = (i % args.profile_every == 0) if args.profile_every > 0 else i >= args.steps - n_samp
        ctx.profile_enable(sample)
        profiled_steps += sample
        wl.step(alone=sample)
    barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    # r05: at N > 1 the host hand-over runs on every rank at once -- SURVEY 8(e): "expected scaling limit = host entropy
    # decode and PCIe, not the kernels" -- and rank 0 reports the slowest rank's step (outside the timed region)
    pcie_all = None
    if world > 1 and not args.headline_only:
        import torch
        legs = {}
        for name, quantised in (("pcie_inclusive", False), ("pcie_inclusive_quantised", True)):
            wl.queues = 2
            barrier()
            r = pcie_pipeline(wl, quantised=quantised)
            t = torch.tensor([r["ms_per_step"], -r["ms_per_step"], r["host_GBs"]], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            r["ms_per_step"] = float(t[0])
            r["fastest_rank_ms_per_step"] = -float(t[1])
            r["Mpix_per_s"] = round(args.frames * world * W * H / (float(t[0]) * 1e-3) / 1e6, 1)
            r["note"] = "max over %d ranks, every rank handing over at once; " % world + r["note"]
            legs[name] = r
        pcie_all = legs

    if rank == 0:
        pictures = args.frames * world * args.steps
        value = pictures * W * H / dt / 1e6
        # dominant kernel = the class with the largest summed event time
        dom = max((k for k in prof if k in ("iiwt_finest", "iiwt_coarse", "obmc", "upsample", "convert")),
                  key=lambda k: prof[k][0])
        samples = args.frames * (W * H * 3 // 2)            # 4:2:0 samples per launch
        # references used per block, from the actual motion field (mode 1/2: one, mode 3: two)
        modes = np.concatenate([m["flags"] & 3 for m in wl.mv_np])
        refs_per_px = float(((modes == 1) | (modes == 2)).mean() + 2 * (modes == 3).mean())
        alg_bytes = {
            "iiwt_finest": 4 * samples,                     # 2 B read + 2 B written per sample
            # levels 1 and 2 are one launch each; the average launch moves (1/4 + 1/16) / 2
            "iiwt_coarse": int(4 * samples * (0.25 + 0.0625) / 2),
            # residual 2 B + output 1 B + 1 B per reference used + 20 B per block.  r04, the combine
            # form: the OBMC launches write the prediction (1 B) and read no residual -- their own algorithmic bytes
            # are 1 B + 1 B per reference used + the vectors; the add lives in the finest wavelet level, which reads
            # 2 B coefficients + 1 B prediction and writes 1 B picture per sample (4 B, the figure it had)
            "obmc": int(((1 if wl.combine else 3) + refs_per_px) * samples) + 20 * args.frames
                    * wl.P["x_num_blocks"] * wl.P["y_num_blocks"],
            # 1 B read + 4 B written per sample of every reference plane of the step (two launches: luma, chroma)
            "upsample": wl.groups * 2 * (W * H * 3 // 2) * 5,
            "convert": 3 * samples,
        }
        # launches of a class per step (the row-per-lane OBMC kernels run luma and chroma planes as
        # two launches: different row lengths, different register budgets); alg_bytes are per STEP
        # for "obmc" and per launch for the others, so every figure below is bytes of the launches
        # of one step / their summed time
        per_step = {k: (n / profiled_steps if profiled_steps else 1) for k, (ms, n) in prof.items()}
        alg_step = dict(alg_bytes)
        for k in ("iiwt_finest", "iiwt_coarse", "convert"):
            alg_step[k] = alg_bytes[k] * per_step.get(k, 1)
        if not prof["iiwt_coarse"][1]:      # r04, the chain form: every level is in the one "iiwt_finest" launch
            alg_step["iiwt_finest"] = int(4 * samples * (1 + 0.25 + 0.0625))
        kernels = {}
        for k, (ms, n) in prof.items():
            if n and k in alg_step:
                step_ms = ms / profiled_steps
                kernels[k] = {"avg_ms": round(ms / n, 4), "launches": n, "ms_per_step": round(step_ms, 4),
                              "alg_GBs": round(alg_step[k] / (step_ms * 1e-3) / 1e9, 1)}
        d_avg = prof[dom][0] / max(prof[dom][1], 1)
        d_step = prof[dom][0] / profiled_steps
        achieved = alg_step[dom] / (d_step * 1e-3) / 1e9
        # the whole 3-level transform (north_star's figure): 4 B per sample of the plane,
        # independent of depth, over the summed time of its launches in one step
        iiwt_ms = (prof["iiwt_finest"][0] + prof["iiwt_coarse"][0]) / profiled_steps
        kernels["iiwt_3_levels"] = {"avg_ms": round(iiwt_ms, 4), "launches": profiled_steps,
                                    "alg_GBs": round(4 * samples / (iiwt_ms * 1e-3) / 1e9, 1),
                                    "frac_of_8TBs": round(4 * samples / (iiwt_ms * 1e-3) / 1e9
                                                          / HBM_PEAK_GBS, 4),
                                    # the read side alone (2 B per sample), the
                                    # quantity rocprof's FETCH_SIZE bounds
                                    "read_frac_of_8TBs": round(2 * samples / (iiwt_ms * 1e-3) / 1e9
                                                               / HBM_PEAK_GBS, 4)}
        # the whole pixel path of a step against SURVEY 8(d)'s algorithmic bytes, whatever the stage structure:
        # wavelet 4 B per sample + OBMC and combine (3 + references used) B per sample + vectors + upsample 5 B per
        # reference sample -- the figure that compares rounds (r03: 971 MB per step in 0.418 ms of kernels)
        path_bytes = 4 * samples + int((3 + refs_per_px) * samples) + 20 * args.frames * wl.P["x_num_blocks"] * wl.P["y_num_blocks"] \
            + alg_bytes["upsample"]
        path_ms = sum(prof[k][0] for k in ("iiwt_finest", "iiwt_coarse", "upsample", "obmc")) / profiled_steps
        pixel_path = {"alg_bytes_per_step": path_bytes, "sum_of_kernel_ms": round(path_ms, 4),
                      "alg_GBs": round(path_bytes / (path_ms * 1e-3) / 1e9, 1),
                      "frac_of_8TBs": round(path_bytes / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "frac_of_8TBs_by_step_time": round(path_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
        # HBM-side bytes per launch from rocprofv3 PMC passes (profiles/, collected offline)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(dom)
        out = {
            "metric": "Mpix/s IIWT+OBMC decode, 2160p s16",
            "value": round(value, 1), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_ms": args.prewarm_ms,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "s16", "data": "synthetic",
            "config": {"workload": "2160p 4:2:0 inter pictures: 3-level DD(9,7) IIWT s16 + "
                       "half-pel upsample of 2 refs + 12x12/8x8 quarter-pel OBMC + add/clamp",
                       "stage_order": "upsample, OBMC prediction, inverse wavelet with the add as its last step (the residual "
                                      "picture stays on the chip)" if wl.combine else "inverse wavelet, upsample, OBMC with the add",
                       "frames_per_step_per_gpu": args.frames, "batches_in_flight": args.queues,
                       "width": W, "height": H,
                       "sharding": "pictures across GPUs, no collective"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "avg_launch_ms": round(d_avg, 4), "launches_per_step": round(per_step[dom], 2),
                         "ms_per_step": round(d_step, 4),
                         "note": "algorithmic bytes of one step's launches of this kernel class / their summed "
                                 "HIP-event time (OBMC: a luma and a chroma launch per step"
                                 + ("; combine form: the launches write the prediction and read no residual -- 1 B + 1 B per "
                                    "reference used per sample + the vectors; see pixel_path for the whole path on SURVEY 8(d)'s bytes)"
                                    if wl.combine else ")")},
            "kernels": kernels,
            "pixel_path": pixel_path,
        }
        if affinity is not None:
            out["host_affinity"] = affinity
        if pcie_all:
            out.update(pcie_all)
        if world == 1 and not args.headline_only:
            # not part of the timed region: the other sizes / views SURVEY 8(d) asks for
            out["iiwt_2160p"] = iiwt_2160p(wl)
            out["iiwt_s32_2160p"] = iiwt_s32_2160p(ctx.device)
            out["iiwt_1080p"] = iiwt_1080p(ctx)
            # ... and with 1 and 32 pictures per launch set (one batch in flight / two): where latency ends
            out["iiwt_1080p"]["pictures_per_launch_set"] = {
                str(n): {"one_batch_ms": r["median_ms"], "one_batch_frac_of_8TBs": r["frac_of_8TBs"],
                         "two_batches_ms": r["two_batches_in_flight"]["ms"],
                         "two_batches_frac_of_8TBs": r["two_batches_in_flight"]["frac_of_8TBs"]}
                for n, r in ((n, iiwt_1080p(ctx, frames=n)) for n in (1, 32))}
            out["kernels"].update(extra_kernels(wl))
            out["pcie_inclusive"] = pcie_pipeline(wl, quantised=False)
            out["frame_layer_2160p"] = frame_layer_2160p()
            if "pipelined_Mpix_per_s" in out["frame_layer_2160p"]:
                out["frame_layer_2160p"]["pipelined_share_of_plane_layer_pcie_inclusive"] = round(
                    out["frame_layer_2160p"]["pipelined_Mpix_per_s"] / out["pcie_inclusive"]["Mpix_per_s"], 3)
            # one batch at a time on one queue, every step timed by itself: median
            wl.queues = 1
            ts = []
            for _ in range(30):
                ctx.timer_begin()
                wl.step()
                ts.append(ctx.timer_end())
            out["one_batch_in_flight"] = {"median_ms_per_step": round(float(np.median(ts)), 4),
                                          "Mpix_per_s": round(args.frames * W * H / float(np.median(ts)) / 1e3, 1)}
        if world == 1 and not args.headline_only:
            wl.queues = args.queues
            out["coherent_motion"] = coherent_motion(wl)
            # r06: the other pictures a decoder meets, each through the whole path: full-pel
            # vectors on plain references (the reference encoder's default: no upsample stage), eighth pel, the 24 / 16
            # block set, and 1080p with 8 and 32 pictures per step
            dev = ctx.device
            out["fullpel_2160p"] = decode_variant(dev, prec=0)
            out["eighthpel_2160p"] = decode_variant(dev, prec=3)
            out["blocks_24_16_2160p"] = decode_variant(dev, xblen=24, xbsep=16)
            # (what the reference's own encoder makes of a 2160p picture by default: 32 x 32 blocks every 16 pixels --
            # schroengine.c:411-453: separation by picture size, full overlap -- and full-pel vectors, schroencoder.c:4488)
            out["encoder_default_2160p"] = decode_variant(dev, xblen=32, xbsep=16, prec=0)
            # (weighted prediction -- a fade: 3, 5 / 2^3; r06: the row kernels' weighted blend.  obmc.hip's general kernel, which
            # such pictures ran until then and which pictures with a gain or a negative weight still run: 2.22 ms of OBMC per step)
            out["weighted_2160p"] = decode_variant(dev, weights=(3, 5, 3))
            out["decode_1080p"] = decode_variant(dev, w=1920, h=1080)
            out["decode_1080p"]["pictures_32_per_step"] = decode_variant(dev, frames=32, w=1920, h=1080, check=False)
            if any(str(out[k].get("parity", "")).startswith("MISMATCH") for k in
                   ("fullpel_2160p", "eighthpel_2160p", "blocks_24_16_2160p", "encoder_default_2160p", "weighted_2160p", "decode_1080p")):
                out["parity_variants"] = "MISMATCH"
        if world == 1 and not args.no_cpu_baseline and not args.headline_only:
            cores = args.cpu_cores or min(16, os.cpu_count() or 1)      # a one-GPU box's CPU share
            v, checked, equal, single = cpu_baseline(wl, cores)
            out["cpu_baseline"] = {"value": round(v, 2), "unit": "Mpix/s", "cores": cores,
                                   "kind": "port", "single_thread": round(single, 2),
                                   "cpu": cpu_model(), "host_cpus": os.cpu_count(),
                                   "sample": "%d pictures (%d threads x 10, one picture per thread at a "
                                   "time) of the same workload + the two reference upsamples, oracle/ C "
                                   "port, gcc -O3" % (10 * cores, cores)}
            # the same on every core this process may use (the 16-thread figure is a one-GPU box's share,
            # not the host's ceiling); capped at 128 threads to bound the sample's memory
            try:
                nproc = len(os.sched_getaffinity(0))
            except AttributeError:
                nproc = os.cpu_count() or 1
            quota = None            # a container's CPU share (cgroup v2 cpu.max "quota period"): more threads than that only queue
            try:
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if q != "max":
                    quota = max(1, -(-int(q) // int(per)))
            except (OSError, ValueError):
                pass
            nthr = max(1, min(nproc, quota or nproc, 128))
            out["cpu_baseline"]["usable_cpus"] = nproc
            out["cpu_baseline"]["cgroup_cpu_quota"] = quota
            if nthr > cores:
                v_all, _, _, _ = cpu_baseline(wl, nthr, reps=4, check=False)
                out["cpu_baseline"]["all_cores"] = {"value": round(v_all, 2), "threads": nthr,
                                                    "sample": "%d pictures (%d threads x 4)" % (4 * nthr, nthr)}
            else:
                out["cpu_baseline"]["all_cores"] = "the %d-thread figure: this process may use %d CPU(s)%s" % (
                    cores, nproc, ", its cgroup's quota is %d" % quota if quota else "")
            want = len(wl.sets) * wl.frames
            out["parity"] = ("bit-exact vs oracle on %d / %d pictures" % (equal, want) if equal == checked == want
                             else "MISMATCH vs oracle: %d of %d pictures equal (%d of %d checked)" % (equal, checked, checked, want))
        if world == 1 and not args.headline_only:
            wl.queues = 2
            out["pcie_inclusive_quantised"] = pcie_pipeline(wl, quantised=True)
            # the two other forms beside it: the r03 form -- the same queues and marks, the host never waits on
            # purpose and spends most of the step blocked inside the two copy calls --, and a batch's copies and kernels in
            # order on one queue (never blocks, the copy engines overlap worse)
            for key, f in (("copy_queues_no_handover_form", "copyq"), ("one_queue_per_batch_form", "queues")):
                alt = pcie_pipeline(wl, quantised=True, form=f)
                out["pcie_inclusive_quantised"][key] = {
                    "ms_per_step": alt["ms_per_step"], "Mpix_per_s": alt["Mpix_per_s"],
                    "host_enqueue_ms_per_step": alt["host_enqueue_ms_per_step"]}
            out["lowdelay_8k"] = lowdelay_8k(ctx)
        print(json.dumps(out))
        if out.get("parity", "").startswith("MISMATCH") or out.get("parity_variants") == "MISMATCH":
            sys.exit(1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
