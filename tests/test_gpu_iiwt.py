"""GPU parity: inverse wavelet (HIP, through the C ABI) vs the CPU oracle.

Mirrors the reference's own GPU parity tests (testsuite/cuda/cuda.c:60-115,
testsuite/opengl/opengl.c:153-330: CPU forward -> upload -> GPU inverse ->
download -> exact compare) and testsuite/wavelet_2d.c's size sweep.
Bit-exact is the bar (integer work).
"""
import numpy as np
import pytest

import oracle_lib as O
import synth

pytestmark = pytest.mark.gpu

FILTERS = range(7)


def gpu_iiwt(ctx, coeffs, depth, filt):
    src = ctx.upload(coeffs)
    dst = ctx.plane(coeffs.shape[0], coeffs.shape[1], coeffs.dtype).fill(0x5a)
    ctx.iiwt_batch([(src, dst)], depth, filt)
    out = dst.download()
    unchanged = src.download()
    src.free()
    dst.free()
    assert np.array_equal(unchanged, coeffs), "source coefficients were modified"
    return out


@pytest.mark.parametrize("dtype", [np.int16, np.int32])
@pytest.mark.parametrize("filt", FILTERS)
def test_one_level_small_sizes(ctx, filt, dtype):
    # every even size class the reference sweeps (wavelet_2d.c:288-299), thinned
    for (h, w) in [(2, 2), (2, 40), (40, 2), (4, 6), (6, 4), (16, 16), (18, 34), (20, 20),
                   (34, 30), (40, 38)]:
        img = synth.image_s(h, w, dtype, seed=h * 41 + w)
        co = O.iwt_2d(img, filt)
        assert np.array_equal(gpu_iiwt(ctx, co, 1, filt), O.iiwt_2d(co, filt)), (filt, h, w)
        fr = synth.full_range(h, w, dtype, seed=h * 7 + w)
        assert np.array_equal(gpu_iiwt(ctx, fr, 1, filt), O.iiwt_2d(fr, filt)), (filt, h, w, "full")


@pytest.mark.parametrize("dtype", [np.int16, np.int32])
@pytest.mark.parametrize("filt", FILTERS)
def test_multi_level(ctx, filt, dtype):
    for (h, w, depth) in [(48, 64, 3), (240, 320, 4), (136, 248, 3), (64, 64, 6), (272, 480, 2)]:
        img = synth.image_s(h, w, dtype, seed=11)
        co = O.forward_iwt(img, depth, filt)
        want = O.inverse_iwt(co, depth, filt)
        got = gpu_iiwt(ctx, co, depth, filt)
        assert np.array_equal(got, want), (filt, h, w, depth)
        assert np.array_equal(got, img), "perfect reconstruction"
        fr = synth.full_range(h, w, dtype, seed=13)
        assert np.array_equal(gpu_iiwt(ctx, fr, depth, filt), O.inverse_iwt(fr, depth, filt))


@pytest.mark.parametrize("filt,depth,h,w", [
    (1, 1, 256, 256),          # BASELINE config 0
    (0, 3, 1080, 1920),        # BASELINE config 1
    (0, 3, 2160, 3840),        # BASELINE config 3 (luma)
    (6, 3, 1080, 1920),
    (5, 2, 1080, 1920),
])
def test_baseline_sizes_s16(ctx, filt, depth, h, w):
    img = synth.image_s(h, w, np.int16, seed=1)
    co = O.forward_iwt(img, depth, filt)
    got = gpu_iiwt(ctx, co, depth, filt)
    assert np.array_equal(got, O.inverse_iwt(co, depth, filt))
    assert np.array_equal(got, img)


@pytest.mark.parametrize("base", ["0", "1"])
@pytest.mark.parametrize("fuse", ["0", "2", "3"])
@pytest.mark.parametrize("dtype", [np.int16, np.int32])
@pytest.mark.parametrize("filt", [0, 1, 2, 3, 4, 6])
def test_fused_levels(ctx, filt, dtype, fuse, base, monkeypatch):
    # SCHRO_HIP_IIWT_FUSE / _FUSE_BASE: levels base .. base+n-1 in one launch (the default
    # fuses the levels above level 0 for s16); same answer whatever the grouping
    monkeypatch.setenv("SCHRO_HIP_IIWT_FUSE", fuse)
    monkeypatch.setenv("SCHRO_HIP_IIWT_FUSE_BASE", base)
    for (h, w, depth) in [(48, 64, 3), (240, 320, 4), (136, 248, 3), (272, 480, 2), (544, 960, 3)]:
        img = synth.image_s(h, w, dtype, seed=17)
        co = O.forward_iwt(img, depth, filt)
        assert np.array_equal(gpu_iiwt(ctx, co, depth, filt), O.inverse_iwt(co, depth, filt)), (h, w, depth)
        fr = synth.full_range(h, w, dtype, seed=19)
        assert np.array_equal(gpu_iiwt(ctx, fr, depth, filt), O.inverse_iwt(fr, depth, filt))


@pytest.mark.parametrize("filt", [0, 1, 2, 3, 4, 6])
@pytest.mark.parametrize("small", ["0", "1"])
def test_register_form_tile_edges(ctx, filt, small, monkeypatch):
    # s16 planes whose sub-band width is a multiple of 4 take the register kernel
    # (iiwt_reg.hip): one wave = 248 columns x (12 - 2H) row pairs.  Sizes around its tile
    # borders: single / several column tiles, last lane inside or outside the picture, the
    # bottom tile moved up (row pairs = k * UR + 1 ...), top and bottom in one tile.
    # SCHRO_HIP_IIWT_SMALL: 12 row pairs per wave (bandwidth-bound levels) or 4 + 2H
    monkeypatch.setenv("SCHRO_HIP_IIWT_SMALL", small)
    for (nr, nc) in [(4, 4), (5, 8), (6, 8), (7, 12), (8, 4), (10, 8), (11, 12), (12, 8), (9, 248), (17, 252), (25, 256), (26, 260), (31, 500), (67, 996)]:
        h, w = 2 * nr, 2 * nc
        fr = synth.full_range(h, w, np.int16, seed=nr * 131 + nc)
        want = O.iiwt_2d(fr, filt)
        assert np.array_equal(gpu_iiwt(ctx, fr, 1, filt), want), (filt, nr, nc)
        monkeypatch.setenv("SCHRO_HIP_IIWT_REG", "0")      # and the LDS kernel agrees
        assert np.array_equal(gpu_iiwt(ctx, fr, 1, filt), want), (filt, nr, nc, "lds")
        monkeypatch.delenv("SCHRO_HIP_IIWT_REG")


def test_baseline_size_s32_haar0(ctx):
    # BASELINE config 5 shape class: s32, Haar (no shift), 3 levels, on one 3840x2160 plane
    # against the oracle, plus the forward -> inverse round trip.  The full 7680x4320 4:2:2
    # picture goes through slices -> wavelet -> v210 in
    # tests/test_gpu_lowdelay.py::test_config5_whole_pixel_path_8k.
    img = synth.image_s(2160, 3840, np.int32, seed=9) * 4   # 10-bit range
    co = O.forward_iwt(img, 3, 3)
    got = gpu_iiwt(ctx, co, 3, 3)
    assert np.array_equal(got, O.inverse_iwt(co, 3, 3))
    assert np.array_equal(got, img)


def test_batch_of_planes_and_strides(ctx):
    # Y,U,V of two pictures in one call, odd strides (unaligned path) and aligned ones
    planes, want = [], []
    for n, (h, w, pad) in enumerate([(144, 176, 0), (72, 88, 6), (72, 88, 0), (144, 176, 2),
                                     (72, 88, 0), (72, 88, 10)]):
        img = synth.image_s(h, w, np.int16, seed=20 + n)
        co = O.forward_iwt(img, 3, 0)
        stride = ((w * 2 + 63) // 64) * 64 + pad
        src = ctx.plane(h, w, np.int16, stride=stride).upload(co)
        dst = ctx.plane(h, w, np.int16, stride=stride + 64)
        planes.append((src, dst))
        want.append(img)
    ctx.iiwt_batch(planes, 3, 0)
    for (src, dst), img in zip(planes, want):
        assert np.array_equal(dst.download(), img)
        src.free()
        dst.free()


def test_argument_errors(ctx):
    import schroedinger_amd as sa
    a = ctx.plane(16, 16, np.int16)
    b = ctx.plane(16, 16, np.int16)
    with pytest.raises(sa.SchroHipError):
        ctx.iiwt_batch([(a, b)], 1, 7)          # filter index out of range
    with pytest.raises(sa.SchroHipError):
        ctx.iiwt_batch([(a, b)], 5, 0)          # 16 is not a multiple of 32
    with pytest.raises(sa.SchroHipError):
        ctx.iiwt_batch([(a, a)], 1, 0)          # in-place is refused, not silently wrong
    a.free()
    b.free()


@pytest.mark.parametrize("filt", [3, 4])
def test_three_level_haar_in_one_pass(ctx, filt, monkeypatch):
    # r03: a depth-3 s32 Haar transform is one launch (iiwt_haar3_s32_kernel) where every plane allows it
    # (width a multiple of 32, 16-byte aligned rows); the per-level form (SCHRO_HIP_IIWT_HAAR3=0) and the
    # oracle must agree with it, also on full-range values (32-bit wrap) and where it does not apply
    for (h, w) in [(8, 32), (72, 96), (264, 480), (48, 40)]:
        for arr in (O.forward_iwt(synth.image_s(h, w, np.int32, seed=h + w) * 4, 3, filt), synth.full_range(h, w, np.int32, seed=3 * h)):
            want = O.inverse_iwt(arr, 3, filt)
            monkeypatch.delenv("SCHRO_HIP_IIWT_HAAR3", raising=False)
            assert np.array_equal(gpu_iiwt(ctx, arr, 3, filt), want), (filt, h, w, "one pass")
            monkeypatch.setenv("SCHRO_HIP_IIWT_HAAR3", "0")
            assert np.array_equal(gpu_iiwt(ctx, arr, 3, filt), want), (filt, h, w, "per level")
    monkeypatch.delenv("SCHRO_HIP_IIWT_HAAR3", raising=False)


@pytest.mark.parametrize("filt", range(7))
@pytest.mark.parametrize("dtype", [np.int16, np.int32])
def test_transform_in_two_calls(ctx, filt, dtype):
    """r04 (SchroHipIwtPlane.ll): the levels above 0 on the level-1 view of the frame into an LL plane of the caller's, then
    level 0 with that plane as its LL band -- together the one call, bit for bit, on the register kernels (s16, large planes),
    the LDS kernels (s32, the fidelity filter, small planes) and for depths 2 .. 4."""
    for (h, w, depth) in ((112, 208, 3), (544, 960, 3), (256, 512, 2), (192, 320, 4)):
        img = synth.image_s(h, w, dtype, seed=7 * filt + depth)
        co = O.forward_iwt(img, depth, filt)
        want = O.inverse_iwt(co, depth, filt)
        d_co = ctx.upload(co)
        whole, split = ctx.plane(h, w, dtype), ctx.plane(h, w, dtype)
        ll = ctx.plane(h // 2, w // 2, dtype)
        ctx.iiwt_batch([(d_co, whole)], depth, filt)
        ctx.iiwt_batch([(d_co.level_view(1), ll)], depth - 1, filt)
        ctx.iiwt_batch([(d_co, split)], 1, filt, ll=[ll])
        got = split.download()
        assert np.array_equal(whole.download(), want), (h, w, depth)
        assert np.array_equal(got, want), (h, w, depth)
        for p in (d_co, whole, split, ll):
            p.free()
