"""GPU parity under random geometry: seeded draws of sizes, strides, filters, depths, block
sets, precisions, weights and slice layouts inside what the reference accepts, each compared
bit for bit with the CPU oracle.  The hand-picked cases of the other test files follow the
reference's own test design; these look for what nobody thought of (tile edges, odd strides,
planes narrower than a tile, batches of unlike planes in one launch)."""
import os

import numpy as np
import pytest

import oracle_lib as O
import schroedinger_amd as sa
import synth
from test_gpu_lowdelay import compare, decode_cpu, decode_gpu
from test_gpu_obmc import run_case


# SCHRO_FUZZ_SCALE multiplies the number of draws, SCHRO_FUZZ_SEED shifts the seeds (a long campaign
# is `SCHRO_FUZZ_SCALE=50 SCHRO_FUZZ_SEED=7 pytest tests/test_gpu_fuzz.py -m gpu`)
SCALE = int(os.environ.get("SCHRO_FUZZ_SCALE", "1"))
SEED = int(os.environ.get("SCHRO_FUZZ_SEED", "0"))
# SCHRO_FUZZ_BIG multiplies the picture sizes of the OBMC / combine draws (more tiles per plane, tile rows that end
# inside blocks, block counts per tile near the kernels' caps); the draws cost BIG^2 as much
BIG = int(os.environ.get("SCHRO_FUZZ_BIG", "1"))

# (tests/conftest.py gives every test six minutes; a campaign's tests get theirs by its size)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(360 + 60 * SCALE * BIG * BIG)]


def test_iiwt_random_batches(ctx):
    rng = np.random.default_rng(101 + SEED)
    for rnd in range(150 * SCALE):
        filt = int(rng.integers(0, 7))
        depth = int(rng.integers(1, 5))
        dtype = [np.int16, np.int32][int(rng.integers(0, 2))]
        planes = []
        for _ in range(int(rng.integers(1, 5))):        # unlike planes in one launch
            unit = 1 << depth
            h, w = unit * int(rng.integers(1, 40)), unit * int(rng.integers(1, 70))
            pad = int(rng.integers(0, 3)) * 8           # source rows wider than the picture
            full = synth.full_range(h, w + pad, dtype, seed=int(rng.integers(1, 1 << 20)))
            if rng.integers(0, 2):
                full = (full >> (3 if dtype == np.int16 else 18)).astype(dtype)
            planes.append((full, w))
        pairs, want = [], []
        for full, w in planes:
            src = ctx.upload(full)
            view = sa.DevicePlane.__new__(sa.DevicePlane)          # the first w columns of src
            view.__dict__.update(src.__dict__)
            view.width = w
            dst = ctx.plane(full.shape[0], w, dtype).fill(0x5a)
            pairs.append((view, dst))
            want.append((O.inverse_iwt(np.ascontiguousarray(full[:, :w]), depth, filt), dst, src))
        ctx.iiwt_batch(pairs, depth, filt)
        for n, (ref, dst, src) in enumerate(want):
            assert np.array_equal(dst.download(), ref), (rnd, n, filt, depth, dtype, ref.shape)
            dst.free()
            src.free()


def test_upsample_and_convert_random(ctx):
    rng = np.random.default_rng(202 + SEED)
    for rnd in range(80 * SCALE):
        h, w = int(rng.integers(1, 150)), int(rng.integers(1, 400))
        pic = synth.picture_u8(h, w, seed=int(rng.integers(1, 1 << 20)), blur=bool(rng.integers(0, 2)))
        src, dst = ctx.upload(pic), ctx.hp_plane(h, w)
        ctx.upsample_batch([(src, dst)])
        up = O.UpComp(pic, upsample=True)
        want = np.zeros((2 * h, 2 * w), np.uint8)
        for k in range(4):
            want[k >> 1::2, k & 1::2] = up.plane(k)
        assert np.array_equal(dst.download(), want), (rnd, h, w)
        src.free()
        dst.free()
        if rnd % 3 == 0:        # r04: the same picture and a second one as a (U, V) pair image
            pv = synth.picture_u8(h, w, seed=int(rng.integers(1, 1 << 20)))
            su, sv, dp = ctx.upload(pic), ctx.upload(pv), ctx.hp_plane(h, w, pair=True)
            ctx.upsample_batch([((su, sv), dp)])
            gu, gv = dp.download()
            upv = O.UpComp(pv, upsample=True)
            want_v = np.zeros((2 * h, 2 * w), np.uint8)
            for k in range(4):
                want_v[k >> 1::2, k & 1::2] = upv.plane(k)
            assert np.array_equal(gu, want) and np.array_equal(gv, want_v), (rnd, h, w, "pair")
            for p in (su, sv, dp):
                p.free()
        dtype = [np.int16, np.int32][rnd & 1]
        ih, iw = h + int(rng.integers(0, 9)), w + int(rng.integers(0, 17))
        res = synth.full_range(ih, iw, dtype, seed=rnd + 5)
        if dtype == np.int32:
            res = res >> 15
        d_res, out = ctx.upload(res), ctx.plane(h, w, np.uint8).fill(7)
        ctx.convert_u8_batch([(d_res, out)])
        assert np.array_equal(out.download(), O.convert_u8(res, w, h)), (rnd, h, w, dtype)
        d_res.free()
        out.free()


def test_obmc_random_geometry(ctx):
    rng = np.random.default_rng(303 + SEED)
    seps = [4, 8, 12, 16, 24, 32]
    for rnd in range(200 * SCALE):
        sep = seps[int(rng.integers(0, len(seps)))]
        blen = sep + 4 * int(rng.integers(0, sep // 4 + 1))
        blen = min(blen, 2 * sep, 64)
        w, h = int(rng.integers(blen, 260 * BIG)), int(rng.integers(blen, 140 * BIG))
        prec = int(rng.integers(0, 4))
        chroma = [(0, 0), (1, 0), (1, 1)][int(rng.integers(0, 3))]
        weights = [(1, 1, 1), (1, 1, 1), (2, 3, 1), (3, 5, 3), (1, 2, 2)][int(rng.integers(0, 5))]
        mv_range = int(rng.integers(1, 120)) << prec
        modes = rng.dirichlet([1, 2, 1, 2])
        run_case(ctx, w, h, blen, sep, prec, weights, chroma, mv_range, seed=int(rng.integers(1, 1 << 16)),
                 res_dtype=[np.int16, np.int32][rnd & 1], modes=tuple(modes))


def test_obmc_random_rectangular_blocks(ctx):
    """r06: block sets drawn independently for x and y (the syntax allows it, schroparams.c:140-205 checks each axis on its
    own): the row kernels' caps count rows by yblen / ybsep and columns by xblen / xbsep."""
    rng = np.random.default_rng(808 + SEED)
    seps = [4, 8, 12, 16, 24, 32]

    def draw():
        sep = seps[int(rng.integers(0, len(seps)))]
        return min(sep + 4 * int(rng.integers(0, sep // 4 + 1)), 2 * sep, 64), sep
    for rnd in range(120 * SCALE):
        (xblen, xbsep), (yblen, ybsep) = draw(), draw()
        w, h = int(rng.integers(xblen, 260 * BIG)), int(rng.integers(yblen, 140 * BIG))
        prec = int(rng.integers(0, 4))
        chroma = [(0, 0), (1, 0), (1, 1)][int(rng.integers(0, 3))]
        weights = [(1, 1, 1), (1, 1, 1), (1, 1, 1), (3, 5, 3), (2, 3, 1)][int(rng.integers(0, 5))]
        try:
            run_case(ctx, w, h, xblen, xbsep, prec, weights, chroma, int(rng.integers(1, 120)) << prec, seed=int(rng.integers(1, 1 << 16)),
                     res_dtype=[np.int16, np.int32][rnd & 1], modes=tuple(rng.dirichlet([1, 2, 1, 2])), pair=bool(rnd % 3 == 0),
                     yblen=yblen, ybsep=ybsep)
        except AssertionError as e:
            raise AssertionError("draw %d: %r" % (rnd, (w, h, xblen, xbsep, yblen, ybsep, prec, chroma, weights))) from e


def test_obmc_random_geometry_pair_images(ctx):
    """r04: the same draws with the chroma planes' references as (U, V) pair images (one UV job per picture)."""
    rng = np.random.default_rng(505 + SEED)
    seps = [4, 8, 12, 16, 24, 32]
    for rnd in range(120 * SCALE):
        sep = seps[int(rng.integers(0, len(seps)))]
        blen = min(sep + 4 * int(rng.integers(0, sep // 4 + 1)), 2 * sep, 64)
        w, h = int(rng.integers(blen, 260 * BIG)), int(rng.integers(blen, 140 * BIG))
        prec = int(rng.integers(1, 4))
        chroma = [(1, 0), (1, 1)][int(rng.integers(0, 2))]
        weights = [(1, 1, 1), (1, 1, 1), (1, 1, 1), (2, 3, 1), (1, 2, 2)][int(rng.integers(0, 5))]
        run_case(ctx, w, h, blen, sep, prec, weights, chroma, int(rng.integers(1, 120)) << prec, seed=int(rng.integers(1, 1 << 16)),
                 res_dtype=[np.int16, np.int32][rnd & 1], modes=tuple(rng.dirichlet([1, 2, 1, 2])), pair=True)


def test_combine_random_geometry(ctx):
    """r04: prediction-only OBMC + the transform's combine step (register epilogue or the scratch route) against the oracle's
    two-step result: random sizes (pictures smaller than the padded transform), filters, depths, sample sizes, block sets."""
    from test_gpu_combine import run_case as combine_case
    rng = np.random.default_rng(606 + SEED)
    seps = [4, 8, 12, 16, 24, 32]       # (r06: 24 and 32 -- the two-segment rows' prediction-only kernels in front of the combine)
    for rnd in range(60 * SCALE):
        sep = seps[int(rng.integers(0, len(seps)))]
        blen = min(sep + 4 * int(rng.integers(0, sep // 4 + 1)), 2 * sep)
        depth = int(rng.integers(1, 5))
        w, h = int(rng.integers(max(blen, 24), 420 * BIG)), int(rng.integers(max(blen, 24), 200 * BIG))
        chroma = [(0, 0), (1, 0), (1, 1)][int(rng.integers(0, 3))]
        args = dict(w=w, h=h, depth=depth, filt=int(rng.integers(0, 7)), dtype=[np.int16, np.int32][int(rng.integers(0, 2))], chroma=chroma,
                    prec=int(rng.integers(0, 4)), blk=(blen, sep), seed=int(rng.integers(1, 1 << 16)), intra=rnd % 7 == 0)
        try:
            combine_case(ctx, **args)
        except AssertionError as e:
            raise AssertionError("draw %d %r: %s" % (rnd, args, e)) from e


def test_iiwt_two_calls_random(ctx):
    """r04 (SchroHipIwtPlane.ll): the levels above 0 on the level-1 views, then level 0 from their LL planes = the one call."""
    rng = np.random.default_rng(707 + SEED)
    for rnd in range(80 * SCALE):
        filt, depth = int(rng.integers(0, 7)), int(rng.integers(2, 5))
        dtype = [np.int16, np.int32][int(rng.integers(0, 2))]
        unit = 1 << depth
        srcs, wholes, splits, lls, want = [], [], [], [], []
        for _ in range(int(rng.integers(1, 4))):
            h, w = unit * int(rng.integers(1, 30)), unit * int(rng.integers(1, 50))
            co = synth.full_range(h, w, dtype, seed=int(rng.integers(1, 1 << 20)))
            co = (co >> (3 if dtype == np.int16 else 18)).astype(dtype)
            srcs.append(ctx.upload(co))
            wholes.append(ctx.plane(h, w, dtype))
            splits.append(ctx.plane(h, w, dtype).fill(0x33))
            lls.append(ctx.plane(h // 2, w // 2, dtype))
            want.append(O.inverse_iwt(co, depth, filt))
        ctx.iiwt_batch(list(zip(srcs, wholes)), depth, filt)
        ctx.iiwt_batch([(s.level_view(1), q) for s, q in zip(srcs, lls)], depth - 1, filt)
        ctx.iiwt_batch(list(zip(srcs, splits)), 1, filt, ll=lls)
        for n in range(len(srcs)):
            assert np.array_equal(wholes[n].download(), want[n]), (rnd, n, filt, depth, dtype)
            assert np.array_equal(splits[n].download(), want[n]), (rnd, n, filt, depth, dtype, "two calls")
        for p in srcs + wholes + splits + lls:
            p.free()


def test_dequant_random_codeblocks(ctx):
    """Core-syntax dequantisation (schro_hip_dequant_batch and, r04, a plan run over the same records): random plane sizes,
    depths, codeblock partitions (down to empty and 1 x 1 codeblocks), zero codeblocks, value widths, both arithmetics."""
    from test_gpu_dequant import D, pack_codeblocks, synthetic_records
    rng = np.random.default_rng(808 + SEED)
    for rnd in range(40 * SCALE):
        dtype, arith = [(np.int16, 0), (np.int16, 1), (np.int32, 0)][int(rng.integers(0, 3))]
        jobs, want, outs = [], [], []
        for _ in range(int(rng.integers(1, 4))):
            depth = int(rng.integers(1, 5))
            unit = 1 << depth
            h, w = unit * int(rng.integers(1, 24)), unit * int(rng.integers(1, 40))
            span = int(rng.choice([60, 3000, 40000]))
            intra = int(rng.integers(0, 2))
            quant = rng.integers(-span, span + 1, (h, w)).astype(np.int32)
            quant[rng.random((h, w)) < rng.uniform(0.2, 0.9)] = 0
            records = synthetic_records(h, w, depth, rng, zero_share=float(rng.uniform(0.0, 0.8)),
                                        counts=(int(rng.integers(1, 9)), int(rng.integers(1, 7))))
            dst = ctx.plane(h, w, dtype).fill(0x5a)
            blob, cbs = pack_codeblocks((h, w), np.dtype(dtype).itemsize, dst.stride, depth, quant, records)
            jobs.append((dst, ctx.upload_bytes(blob), cbs, intra))
            ref = np.full((h, w), 0x5a5a5a5a & (0xffff if dtype == np.int16 else 0xffffffff), np.uint32).astype(dtype)
            for (index, x0, y0, x1, y1, zero, qi) in records:
                band, qb = D.subband_view(ref, depth, index), D.subband_view(quant, depth, index)
                if x1 > x0 and y1 > y0:
                    O.dequant_codeblock(band[y0:y1, x0:x1], None if zero else qb[y0:y1, x0:x1], qi, intra, arith)
            want.append(ref)
            outs.append(dst)
        ctx.dequant_batch(jobs, arith)
        for n, (dst, ref) in enumerate(zip(outs, want)):
            assert np.array_equal(dst.download(), ref), (rnd, n, dtype, arith, "batch")
            dst.fill(0x5a)
        plan = ctx.dequant_plan(jobs, arith)
        plan.run(jobs)
        for n, (dst, ref) in enumerate(zip(outs, want)):
            assert np.array_equal(dst.download(), ref), (rnd, n, dtype, arith, "plan")
        plan.free()
        for dst, dev, _, _ in jobs:
            dst.free()
            dev.free()


def test_frame_layer_random_pictures(ctx):
    """The SchroFrame-shaped stage calls on random pictures (sizes that are not multiples of anything, 4:4:4 / 4:2:2 /
    4:2:0, every filter, depths 1 .. 4, block sets, precisions): the reference's stage order (inverse transform, render with
    add = TRUE) and the r04 order (render add = FALSE, combine transform) both give the oracle's picture."""
    import ctypes as C
    from schroedinger_amd import _lib, frames
    lib = ctx.lib
    rng = np.random.default_rng(909 + SEED)
    seps = [4, 8, 12, 16]
    for rnd in range(30 * SCALE):
        hs, vs = [(0, 0), (1, 0), (1, 1)][int(rng.integers(0, 3))]
        sep = seps[int(rng.integers(0, len(seps)))]
        blen = min(sep + 4 * int(rng.integers(0, sep // 4 + 1)), 2 * sep)
        depth, filt, prec = int(rng.integers(1, 5)), int(rng.integers(0, 7)), int(rng.integers(0, 4))
        w = 2 * int(rng.integers(max(blen, 16) // 2 + 1, 200 * BIG))       # (even sizes: the chroma formats' own rule)
        h = 2 * int(rng.integers(max(blen, 16) // 2 + 1, 120 * BIG))
        pd = [(h, w), (-(-h >> vs), -(-w >> hs)), (-(-h >> vs), -(-w >> hs))]
        unit = 1 << depth
        iw = [(-(-ph // unit) * unit, -(-pw // unit) * unit) for (ph, pw) in pd]
        P = synth.motion_params(w, h, blen, sep, prec, (1, 1, 1), (hs, vs))
        params = frames.make_params(wavelet_filter_index=filt, transform_depth=depth, iwt_luma_width=iw[0][1], iwt_luma_height=iw[0][0],
                                    iwt_chroma_width=iw[1][1], iwt_chroma_height=iw[1][0], num_refs=2,
                                    **{k: P[k] for k in ("xblen_luma", "yblen_luma", "xbsep_luma", "ybsep_luma", "mv_precision",
                                                         "picture_weight_bits", "picture_weight_1", "picture_weight_2", "x_num_blocks", "y_num_blocks")})
        mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], int(rng.integers(1, 60)) << prec, seed=int(rng.integers(1, 1 << 16)))
        resid = [synth.image_s(ih, iwd, np.int16, seed=int(rng.integers(1, 1 << 16))) for (ih, iwd) in iw]
        coeffs = [O.forward_iwt(r, depth, filt) for r in resid]
        res_want = [O.inverse_iwt(c, depth, filt) for c in coeffs]
        fmt16, fmt8 = frames.frame_format(np.int16, hs, vs), frames.frame_format(np.uint8, hs, vs)
        # (picture->transform_frame and picture->frame as the reference allocates them: schro_video_format_get_iwt_alloc_size
        # rounds the luma size up so far that the chroma components hold their own padded transform too)
        fw, fh = -(-w // (unit << hs)) * (unit << hs), -(-h // (unit << vs)) * (unit << vs)
        padded = []
        for k, c in enumerate(coeffs):
            a = np.zeros((fh >> (vs if k else 0), fw >> (hs if k else 0)), np.int16)
            a[:c.shape[0], :c.shape[1]] = c
            padded.append(a)
        transform_frame = frames.HostFrame(padded, hs, vs)
        refs_np = [[synth.picture_u8(ph, pw, seed=int(rng.integers(1, 1 << 16))) for (ph, pw) in pd] for r in range(2)]
        refs, held = [], []
        for r in range(2):
            d = frames.DeviceFrame(ctx, fmt8, w, h).upload(frames.HostFrame(refs_np[r], hs, vs))
            held.append(d)
            if prec > 0:
                u = frames.DeviceFrame(ctx, fmt8, w, h, upsampled=True)
                sa.check(lib.schro_upsampled_hipframe_upsample(u.ptr(), d.ptr()))
                held.append(u)
                refs.append(u)
            else:
                refs.append(d)
        want = [O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(refs_np[0][k], upsample=prec > 0),
                                O.UpComp(refs_np[1][k], upsample=prec > 0), res_want[k], pw, ph) for k, (ph, pw) in enumerate(pd)]
        motion = _lib.Motion(refs[0].ptr(), refs[1].ptr(), mv.ctypes.data, C.pointer(params))
        tag = (rnd, w, h, (hs, vs), depth, filt, prec, (blen, sep))
        # the reference's order
        frame, out = frames.DeviceFrame(ctx, fmt16, fw, fh), frames.DeviceFrame(ctx, fmt8, w, h)
        sa.check(lib.schro_frame_inverse_iwt_transform_hip(frame.ptr(), transform_frame.ptr(), C.byref(params)))
        sa.check(lib.schro_motion_render_hip(C.byref(motion), None, frame.ptr(), 1, out.ptr()))
        for k, g in enumerate(out.download()):
            assert np.array_equal(g, want[k]), tag + (k, "stage order of the reference")
        # r04: prediction first, the transform adds it
        mc_tmp, out2 = frames.DeviceFrame(ctx, fmt8, w, h), frames.DeviceFrame(ctx, fmt8, w, h)
        sa.check(lib.schro_motion_render_hip(C.byref(motion), mc_tmp.ptr(), None, 0, None))
        sa.check(lib.schro_frame_inverse_iwt_transform_combine_hip(out2.ptr(), transform_frame.ptr(), C.byref(params), mc_tmp.ptr()))
        for k, g in enumerate(out2.download()):
            assert np.array_equal(g, want[k]), tag + (k, "combine order")
        for f in [frame, out, mc_tmp, out2] + held:
            f.unref()


def test_pack_random_sizes(ctx):
    """Packed copy-out of u8 pictures (YUYV / UYVY / AYUV): random source sizes and chroma formats, destinations smaller
    (crop) and larger (edge extension) than the source, several unlike pictures per launch."""
    from test_gpu_pack import gpu_pack, planes
    rng = np.random.default_rng(1010 + SEED)
    fmts = [sa.FORMAT_YUYV, sa.FORMAT_UYVY, sa.FORMAT_AYUV]
    for rnd in range(40 * SCALE):
        cases = []
        for _ in range(int(rng.integers(1, 6))):
            hs, vs = [(0, 0), (1, 0), (1, 1)][int(rng.integers(0, 3))]
            w, h = int(rng.integers(1, 300)), int(rng.integers(1, 90))
            pl = planes(w, h, hs, vs, seed=int(rng.integers(1, 1 << 16)))
            # (crop OR extension: a destination narrower and taller than its source, or the reverse, is refused -- no decoder has one)
            sign = 1 if rng.integers(0, 2) else -1
            W, H = max(1, w + sign * int(rng.integers(0, 9))), max(1, h + sign * int(rng.integers(0, 6)))
            cases.append((pl, hs, vs, fmts[int(rng.integers(0, 3))], W, H))
        for (pl, hs, vs, f, W, H), g in zip(cases, gpu_pack(ctx, cases)):
            assert np.array_equal(g, O.pack_u8(pl, hs, vs, f, W, H)), (rnd, f, (hs, vs), pl[0].shape, W, H)


def test_lowdelay_random_layouts(ctx):
    rng = np.random.default_rng(404 + SEED)
    for rnd in range(150 * SCALE):
        depth = int(rng.integers(0, 5))
        unit = 1 << depth
        chroma = [(0, 0), (1, 0), (1, 1)][int(rng.integers(0, 3))]
        w = unit * (2 if chroma[0] else 1) * int(rng.integers(1, 24))
        h = unit * (2 if chroma[1] else 1) * int(rng.integers(1, 12))
        bpp = [2, 4][int(rng.integers(0, 2))]
        P = synth.lowdelay_params(w, h, chroma, depth, 8, 8, 1)
        P["n_horiz_slices"] = int(rng.integers(1, max(2, P["iwt_luma_width"] // 4)))
        P["n_vert_slices"] = int(rng.integers(1, max(2, P["iwt_luma_height"] // 4)))
        samples = (P["iwt_luma_width"] * P["iwt_luma_height"] + 2 * P["iwt_chroma_width"] * P["iwt_chroma_height"])
        per_slice = samples / (P["n_horiz_slices"] * P["n_vert_slices"])
        den = int(rng.integers(1, 5))
        P["slice_bytes_num"] = max(den * 4, int(den * (4 + per_slice * rng.uniform(0.15, 0.6))))
        P["slice_bytes_denom"] = den
        P["quant_matrix"] = [int(v) for v in rng.integers(0, 12, 1 + 3 * depth)]
        q = synth.quantised_planes(P, seed=int(rng.integers(1, 1 << 16)), scale=float(rng.uniform(0.4, 2.5)),
                                   big_every=int(rng.integers(0, 2)) * 37, big_range=1 << int(rng.integers(8, 31)))
        bi = synth.lowdelay_base_index(P, seed=rnd, lo=0, hi=int(rng.integers(0, 128)))
        data = O.lowdelay_write(q, P, bpp, bi, pad_bit=rnd & 1, y_length_bias=int(rng.integers(-9, 10)) * (rnd % 3 == 0))
        got = decode_gpu(ctx, [data], P, bpp, misalign=rnd % 4)
        compare(got[0], decode_cpu(data, P, bpp), "layout %d %s" % (rnd, P))
