"""GPU parity: packed copy-out (schro_hip_pack_u8_batch) vs the CPU oracle
(schro_frame_convert with a YUYV / UYVY / AYUV destination, schroframe.c:869-979)."""
import numpy as np
import pytest

import oracle_lib as O
import schroedinger_amd as sa
import synth

pytestmark = pytest.mark.gpu


def planes(w, h, hs, vs, seed):
    cw, ch = -(-w // (1 << hs)), -(-h // (1 << vs))
    return [synth.picture_u8(h, w, seed=seed), synth.picture_u8(ch, cw, seed=seed + 1),
            synth.picture_u8(ch, cw, seed=seed + 2)]


def gpu_pack(ctx, cases):
    jobs, outs = [], []
    for (pl, hs, vs, fmt, W, H) in cases:
        dev = [ctx.upload(p) for p in pl]
        groups = W if fmt == sa.FORMAT_AYUV else W // 2
        dst = ctx.plane(H, max(4 * groups, 4), np.uint8).fill(0x5a)
        jobs.append((dev, hs, vs, dst, W, H, fmt))
        outs.append((dst, groups, dev))
    ctx.pack_u8_batch(jobs)
    res = []
    for dst, groups, dev in outs:
        res.append(dst.download()[:, :4 * groups])
        dst.free()
        for d in dev:
            d.free()
    return res


@pytest.mark.parametrize("fmt", [sa.FORMAT_YUYV, sa.FORMAT_UYVY, sa.FORMAT_AYUV])
@pytest.mark.parametrize("chroma", [(0, 0), (1, 0), (1, 1)])
def test_pack_sizes_crop_and_extend(ctx, fmt, chroma):
    hs, vs = chroma
    cases = []
    for (w, h) in [(16, 8), (17, 9), (34, 20), (2, 2), (1, 1), (64, 48), (200, 37), (1030, 5)]:
        pl = planes(w, h, hs, vs, seed=w * 7 + h)
        for (W, H) in [(w, h), (w + 5, h + 3), (w + 1, h), (max(w - 3, 1), max(h - 2, 1)), (max(w - 1, 1), h)]:
            cases.append((pl, hs, vs, fmt, W, H))
    got = gpu_pack(ctx, cases)              # one launch for all of them
    for (pl, hs, vs, f, W, H), g in zip(cases, got):
        assert np.array_equal(g, O.pack_u8(pl, hs, vs, f, W, H)), (f, chroma, pl[0].shape, W, H)


def test_pack_picture_sizes(ctx):
    # decoded 1080p / 2160p 4:2:0 pictures to the packed formats an application asks for
    for (w, h) in [(1920, 1080), (3840, 2160)]:
        pl = planes(w, h, 1, 1, seed=11)
        cases = [(pl, 1, 1, fmt, w, h) for fmt in (sa.FORMAT_YUYV, sa.FORMAT_UYVY, sa.FORMAT_AYUV)]
        for (c, g) in zip(cases, gpu_pack(ctx, cases)):
            assert np.array_equal(g, O.pack_u8(pl, 1, 1, c[3], w, h)), (c[3], w, h)


def test_pack_rows_outside_the_picture_are_untouched(ctx):
    pl = planes(40, 12, 1, 1, seed=5)
    dev = [ctx.upload(p) for p in pl]
    dst = ctx.plane(16, 96, np.uint8).fill(0x5a)        # 4 rows and 16 bytes per row to spare
    ctx.pack_u8_batch([(dev, 1, 1, dst, 40, 12, sa.FORMAT_UYVY)])
    out = dst.download()
    assert np.array_equal(out[:12, :80], O.pack_u8(pl, 1, 1, O.FORMAT_UYVY, 40, 12))
    assert (out[12:] == 0x5a).all() and (out[:, 80:] == 0x5a).all()


def test_pack_argument_errors(ctx):
    pl = planes(16, 8, 1, 1, seed=1)
    dev = [ctx.upload(p) for p in pl]
    dst = ctx.plane(16, 128, np.uint8)
    with pytest.raises(sa.SchroHipError):
        ctx.pack_u8_batch([(dev, 1, 1, dst, 20, 4, sa.FORMAT_YUYV)])     # wider and shorter
    with pytest.raises(sa.SchroHipError):
        ctx.pack_u8_batch([(dev, 1, 1, dst, 16, 8, 0x103)])              # not a format of this call
    with pytest.raises(sa.SchroHipError):
        ctx.pack_u8_batch([(dev, 0, 1, dst, 16, 8, sa.FORMAT_YUYV)])     # 4:4:0 does not exist


def signed_planes(w, h, dtype, seed):
    cw = -(-w // 2)
    span = 1500 if dtype == np.int16 else 70000
    mk = lambda hh, ww, sd: ((synth.lcg(hh * ww, sd).astype(np.int64) % (2 * span)) - span).reshape(hh, ww).astype(dtype)
    return [mk(h, w, seed), mk(h, cw, seed + 1), mk(h, cw, seed + 2)]


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.int32])
def test_pack_v210(ctx, dtype):
    # the copy-out of > 8-bit streams (s16 / s32 4:2:2 frames, e.g. the 10-bit 8K 4:2:2
    # configuration) and of 8-bit pictures into v210
    cases = []
    for (w, h) in [(12, 4), (13, 3), (6, 2), (1, 1), (50, 9), (96, 16), (1920, 8), (3842, 3)]:
        for (hs, vs) in ([(0, 0), (1, 0), (1, 1)] if dtype == np.uint8 else [(1, 0)]):
            pl = planes(w, h, hs, vs, seed=w + h) if dtype == np.uint8 else signed_planes(w, h, dtype, w + h)
            for (W, H) in [(w, h), (w + 7, h + 2), (max(w - 5, 1), max(h - 1, 1))]:
                cases.append((pl, hs, vs, W, H))
    jobs, outs = [], []
    for (pl, hs, vs, W, H) in cases:
        dev = [ctx.upload(p) for p in pl]
        dst = ctx.plane(H, 16 * (-(-W // 6)), np.uint8).fill(0x5a)
        jobs.append((dev, hs, vs, dst, W, H))
        outs.append((dst, dev))
    ctx.pack_v210_batch(jobs)
    for (pl, hs, vs, W, H), (dst, dev) in zip(cases, outs):
        assert np.array_equal(dst.download(), O.pack_v210(pl, hs, vs, W, H)), (dtype, pl[0].shape, hs, vs, W, H)
        dst.free()
        for d in dev:
            d.free()


def test_pack_v210_refuses_non_422_signed_sources(ctx):
    pl = signed_planes(16, 4, np.int16, 1)
    dev = [ctx.upload(p) for p in pl]
    dst = ctx.plane(4, 48, np.uint8)
    with pytest.raises(sa.SchroHipError):
        ctx.pack_v210_batch([(dev, 0, 0, dst, 16, 4)])


# ---- v216 / ARGB / AY64 and the > 8-bit output shift --------------------------
WIDE = [(sa.FORMAT_V216, 1, 0, lambda w: 8 * (w // 2)), (sa.FORMAT_ARGB, 0, 0, lambda w: 4 * w),
        (sa.FORMAT_AY64, 0, 0, lambda w: 8 * w)]


def wide_planes(w, h, hs, dtype, seed):
    cw = -(-w >> hs)
    if dtype == np.uint8:
        mk = lambda hh, ww, sd: (synth.lcg(hh * ww, sd) & 0xff).reshape(hh, ww).astype(np.uint8)
    else:
        span = 40000 if dtype == np.int32 else 32768        # AY64 clamps; s32 -> s16 truncates
        mk = lambda hh, ww, sd: ((synth.lcg(hh * ww, sd).astype(np.int64) * 7919 % (2 * span)) - span).reshape(hh, ww).astype(dtype)
    return [mk(h, w, seed), mk(h, cw, seed + 1), mk(h, cw, seed + 2)]


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.int32])
@pytest.mark.parametrize("fmt,hs,vs,row_bytes", WIDE)
def test_pack_wide_formats(ctx, fmt, hs, vs, row_bytes, dtype):
    cases = []
    for (w, h) in [(12, 4), (13, 3), (2, 2), (1, 1), (50, 9), (96, 16), (1920, 8), (3842, 3)]:
        pl = wide_planes(w, h, hs, dtype, w + h)
        for (W, H) in [(w, h), (w + 7, h + 2), (max(w - 5, 1), max(h - 1, 1))]:
            if row_bytes(W) > 0:
                cases.append((pl, W, H))
    jobs, outs = [], []
    for (pl, W, H) in cases:
        dev = [ctx.upload(p) for p in pl]
        dst = ctx.plane(H, row_bytes(W), np.uint8).fill(0x5a)
        jobs.append((dev, hs, vs, dst, W, H, fmt))
        outs.append((dst, dev))
    ctx.pack_wide_batch(jobs)
    for (pl, W, H), (dst, dev) in zip(cases, outs):
        assert np.array_equal(dst.download(), O.pack_wide(pl, hs, vs, W, H, fmt)), (hex(fmt), dtype, pl[0].shape, W, H)
        dst.free()
        for d in dev:
            d.free()


def test_pack_wide_refuses_other_chroma_formats(ctx):
    pl = [ctx.upload(p) for p in wide_planes(16, 4, 1, np.int16, 1)]
    dst = ctx.plane(4, 64, np.uint8)
    with pytest.raises(sa.SchroHipError, match="4:4:4"):
        ctx.pack_wide_batch([(pl, 1, 0, dst, 16, 4, sa.FORMAT_ARGB)])
    with pytest.raises(sa.SchroHipError, match="4:2:2"):
        ctx.pack_wide_batch([(pl, 0, 0, dst, 16, 4, sa.FORMAT_V216)])


@pytest.mark.parametrize("dtype", [np.int16, np.int32])
def test_shift_right(ctx, dtype):
    # schro_frame_shift_right: the intra frame of a > 8-bit stream on its way to an 8-bit picture
    planes = [synth.full_range(h, w, dtype, seed=3 + n) for n, (h, w) in enumerate([(1, 1), (7, 13), (64, 520), (270, 481)])]
    for shift in (1, 2, 4):
        dev = [ctx.upload(p) for p in planes]
        ctx.shift_right_batch(dev, shift)
        for d, p in zip(dev, planes):
            assert np.array_equal(d.download(), O.shift_right(p, shift)), (dtype, shift, p.shape)
            d.free()
