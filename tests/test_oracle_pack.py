"""The packed copy-out oracle (oracle/oracle_pack.c) against the reference's own packer
kernel and against the reference's line-rendering chain restated in numpy
(schroframe.c:869-979, schrovirtframe.c:943-991, 1230-1247, 1438-1537, 1823-1895)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import synth


def planes(w, h, hs, vs, seed):
    cw, ch = -(-w // (1 << hs)), -(-h // (1 << vs))
    return [synth.picture_u8(h, w, seed=seed), synth.picture_u8(ch, cw, seed=seed + 1),
            synth.picture_u8(ch, cw, seed=seed + 2)]


def chain_numpy(pl, hs, vs, fmt, W, H):
    """The reference's virtual-frame chain, stage by stage on whole arrays."""
    y, u, v = [p.astype(np.uint8) for p in pl]
    h, w = y.shape
    t_hs = 0 if fmt == O.FORMAT_AYUV else 1

    def subsample(c):                   # convert_4xx_4yy: to (t_hs, v_shift 0)
        tw = -(-w // (1 << t_hs))
        rows = np.arange(h) >> 1 if vs else np.arange(h)
        if t_hs == hs:
            cols = np.arange(tw)
        elif t_hs > hs:
            cols = 2 * np.arange(tw)
        else:
            cols = np.arange(tw) >> 1
        return c[rows][:, cols]

    def fit(c, cw, ch):                 # crop_u8 / edge_extend_u8
        rows = np.minimum(np.arange(ch), c.shape[0] - 1)
        cols = np.minimum(np.arange(cw), c.shape[1] - 1)
        return c[rows][:, cols]

    cw = -(-W // (1 << t_hs))
    Y, U, V = fit(y, W, H), fit(subsample(u), cw, H), fit(subsample(v), cw, H)
    if fmt == O.FORMAT_AYUV:
        out = np.empty((H, W, 4), np.uint8)
        out[..., 0], out[..., 1], out[..., 2], out[..., 3] = 0xff, Y, U, V
        return out.reshape(H, 4 * W)
    n = W // 2
    out = np.empty((H, n, 4), np.uint8)
    y0, y1 = Y[:, 0:2 * n:2], Y[:, 1:2 * n:2]
    if fmt == O.FORMAT_YUYV:
        out[..., 0], out[..., 1], out[..., 2], out[..., 3] = y0, U[:, :n], y1, V[:, :n]
    else:
        out[..., 0], out[..., 1], out[..., 2], out[..., 3] = U[:, :n], y0, V[:, :n], y1
    return out.reshape(H, 4 * n)


@pytest.mark.parametrize("fmt", [O.FORMAT_YUYV, O.FORMAT_UYVY, O.FORMAT_AYUV])
@pytest.mark.parametrize("chroma", [(0, 0), (1, 0), (1, 1)])
def test_pack_matches_the_chain(fmt, chroma):
    hs, vs = chroma
    for (w, h) in [(16, 8), (17, 9), (34, 20), (2, 2), (1, 1), (64, 48)]:
        pl = planes(w, h, hs, vs, seed=w * 7 + h)
        for (W, H) in [(w, h), (w + 5, h + 3), (w + 1, h), (max(w - 3, 1), max(h - 2, 1)), (max(w - 1, 1), h)]:
            got = O.pack_u8(pl, hs, vs, fmt, W, H)
            assert np.array_equal(got, chain_numpy(pl, hs, vs, fmt, W, H)), (fmt, chroma, w, h, W, H)


@pytest.mark.skipif(not O.ref_available(), reason="reference kernels not built (oracle/_ref)")
def test_yuyv_byte_order_is_the_reference_kernels():
    # orc_packyuyv (schroorc.orc:718-734) compiled from the reference's schroorc-dist.c
    w, h = 64, 4
    pl = planes(w, h, 1, 0, seed=3)
    want = O.pack_u8(pl, 1, 0, O.FORMAT_YUYV, w, h)
    R = O.reforc()
    R.orc_packyuyv.argtypes = [C.c_void_p] * 4 + [C.c_int]
    for i in range(h):
        row = np.zeros(2 * w, np.uint8)
        y, u, v = [np.ascontiguousarray(p[i]) for p in pl]
        R.orc_packyuyv(row.ctypes.data, y.ctypes.data, u.ctypes.data, v.ctypes.data, w // 2)
        assert np.array_equal(row, want[i])


def test_pack_refuses_mixed_crop_and_extension():
    pl = planes(16, 8, 1, 1, seed=1)
    src = O.PackSrc()
    for k in range(3):
        src.data[k] = pl[k].ctypes.data
        src.stride[k] = pl[k].strides[0]
    src.width, src.height, src.h_shift, src.v_shift = 16, 8, 1, 1
    out = np.zeros((16, 64), np.uint8)
    assert O.lib().oracle_pack_u8(out.ctypes.data, 64, O.FORMAT_YUYV, 20, 4, C.byref(src)) == -1


def v210_numpy(pl, hs, vs, W, H):
    """pack_v210 / pack_v210_s16 (schrovirtframe.c:1044-1210) on whole arrays."""
    dt = pl[0].dtype
    y, u, v = pl
    h, w = y.shape
    if dt == np.uint8:
        rows = np.arange(h) >> 1 if vs else np.arange(h)
        tw = -(-w // 2)
        cols = np.arange(tw) if hs == 1 else 2 * np.arange(tw)
        u, v = u[rows][:, cols], v[rows][:, cols]

    def fit(c, cw, ch):
        return c[np.minimum(np.arange(ch), c.shape[0] - 1)][:, np.minimum(np.arange(cw), c.shape[1] - 1)]

    ng = -(-W // 6)
    Y = fit(y, 6 * ng, H).astype(np.int64)
    U, V = fit(u, 3 * ng, H).astype(np.int64), fit(v, 3 * ng, H).astype(np.int64)
    if dt == np.uint8:
        to10 = lambda a: (a << 2) | (a >> 6)
    else:
        to10 = lambda a: np.clip(a.astype(np.int16).astype(np.int64) + 512, 0, 1023)
    Y, U, V = to10(Y), to10(U), to10(V)
    Y[:, W:] = 0
    cin = (2 * np.arange(3 * ng)) < W
    U[:, ~cin] = 0
    V[:, ~cin] = 0
    Y, U, V = Y.reshape(H, ng, 6), U.reshape(H, ng, 3), V.reshape(H, ng, 3)
    words = np.empty((H, ng, 4), np.uint32)
    words[..., 0] = (V[..., 0] << 20) | (Y[..., 0] << 10) | U[..., 0]
    words[..., 1] = (Y[..., 2] << 20) | (U[..., 1] << 10) | Y[..., 1]
    words[..., 2] = (U[..., 2] << 20) | (Y[..., 3] << 10) | V[..., 1]
    words[..., 3] = (Y[..., 5] << 20) | (V[..., 2] << 10) | Y[..., 4]
    return words.astype("<u4").view(np.uint8).reshape(H, 16 * ng)


def signed_planes(w, h, dtype, seed):
    cw = -(-w // 2)
    # mostly inside the 10-bit range, some samples beyond it (clamp) and, for s32, beyond 16
    # bits (the truncation of convert_s16_s32)
    span = 1500 if dtype == np.int16 else 70000
    mk = lambda hh, ww, sd: ((synth.lcg(hh * ww, sd).astype(np.int64) % (2 * span)) - span).reshape(hh, ww).astype(dtype)
    return [mk(h, w, seed), mk(h, cw, seed + 1), mk(h, cw, seed + 2)]


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.int32])
def test_v210_matches_the_chain(dtype):
    for (w, h) in [(12, 4), (13, 3), (6, 2), (1, 1), (50, 9), (96, 16)]:
        for (hs, vs) in ([(0, 0), (1, 0), (1, 1)] if dtype == np.uint8 else [(1, 0)]):
            pl = planes(w, h, hs, vs, seed=w + h) if dtype == np.uint8 else signed_planes(w, h, dtype, w + h)
            for (W, H) in [(w, h), (w + 7, h + 2), (max(w - 5, 1), max(h - 1, 1))]:
                got = O.pack_v210(pl, hs, vs, W, H)
                assert np.array_equal(got, v210_numpy(pl, hs, vs, W, H)), (dtype, w, h, hs, vs, W, H)


# ---- v216 / ARGB / AY64 and the > 8-bit output shift --------------------------
def test_shift_right_against_the_compiled_reference_kernels():
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import ctypes as C
    ref = O.reforc()
    v16 = np.arange(-32768, 32768, dtype=np.int32).astype(np.int16).reshape(256, 256)
    rng = np.random.default_rng(1)
    v32 = rng.integers(-2 ** 31, 2 ** 31, (64, 200)).astype(np.int32)
    for shift in (1, 2, 4, 6, 8):
        a = v16.copy()
        for row in a:
            ref.orc_add_const_rshift_s16(row.ctypes.data_as(C.c_void_p), (1 << shift) >> 1, shift, row.size)
        assert np.array_equal(O.shift_right(v16, shift), a), shift
        b = v32.copy()
        for row in b:
            ref.orc_add_const_rshift_s32(row.ctypes.data_as(C.c_void_p), (1 << shift) >> 1, shift, row.size)
        assert np.array_equal(O.shift_right(v32, shift), b), shift


def test_depth_conversions_against_the_compiled_reference_kernels():
    # what feeds pack_v216 / pack_argb / pack_ayuv64: x - 128 from u8, truncation from s32
    if not O.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import ctypes as C
    ref = O.reforc()
    u8 = np.arange(256, dtype=np.uint8)
    s16 = np.zeros(256, np.int16)
    ref.orc_offsetconvert_s16_u8(s16.ctypes.data_as(C.c_void_p), u8.ctypes.data_as(C.c_void_p), 256)
    assert np.array_equal(s16, u8.astype(np.int16) - 128)
    s32 = np.array([0, 1, -1, 32767, 32768, -32768, -32769, 65536 + 5, -70000, 2 ** 31 - 1], np.int32)
    t16 = np.zeros(s32.size, np.int16)
    ref.orc_convert_s16_s32(t16.ctypes.data_as(C.c_void_p), s32.ctypes.data_as(C.c_void_p), s32.size)
    assert np.array_equal(t16, s32.astype(np.int16))
    # the oracle applies exactly these before packing: ARGB of (y, 0, 0) is (0xff, y, y, y) mod 256
    y = (u8.astype(np.int16) - 128).reshape(1, 256)
    z = np.zeros((1, 256), np.int16)
    argb = O.pack_wide([y, z, z], 0, 0, 256, 1, 0x103).reshape(256, 4)
    assert np.array_equal(argb[:, 0], np.full(256, 0xff, np.uint8))
    for c in (1, 2, 3):
        assert np.array_equal(argb[:, c], (u8.astype(np.int16) - 128).astype(np.uint8))
    argb8 = O.pack_wide([u8.reshape(1, 256), np.full((1, 256), 128, np.uint8), np.full((1, 256), 128, np.uint8)],
                        0, 0, 256, 1, 0x103)
    assert np.array_equal(argb8, argb.reshape(1, -1))


def test_wide_packs_against_a_numpy_model():
    rng = np.random.default_rng(7)
    h, w = 9, 22
    # AY64 from s32 4:4:4
    pl = [rng.integers(-70000, 70000, (h, w)).astype(np.int32) for _ in range(3)]
    got = O.pack_wide(pl, 0, 0, w, h, 0x107).view("<u2").reshape(h, w, 4)
    assert np.all(got[:, :, 0] == 0xffff)
    for k in range(3):
        assert np.array_equal(got[:, :, 1 + k], np.clip(pl[k].astype(np.int64) + 0x8000, 0, 0xffff))
    # ARGB from s16 4:4:4: the YCoCg-R inverse
    pl = [rng.integers(-300, 300, (h, w)).astype(np.int16) for _ in range(3)]
    y, co, cg = [p.astype(np.int64) for p in pl]
    t = y + (cg >> 1)
    b = t - (co >> 1)
    got = O.pack_wide(pl, 0, 0, w, h, 0x103).reshape(h, w, 4)
    assert np.array_equal(got[:, :, 1], (b + co).astype(np.uint8)) and np.array_equal(got[:, :, 2], (t + cg).astype(np.uint8))
    assert np.array_equal(got[:, :, 3], b.astype(np.uint8))
    # v216 from s16 4:2:2: bytes of the little-endian s16 lines, each doubled
    yp = rng.integers(-512, 512, (h, w)).astype(np.int16)
    up, vp = [rng.integers(-512, 512, (h, w // 2)).astype(np.int16) for _ in range(2)]
    got = O.pack_wide([yp, up, vp], 1, 0, w, h, 0x105).reshape(h, w // 2, 8)
    yb, ub, vb = [p.view(np.uint8) for p in (yp, up, vp)]
    j = np.arange(w // 2)
    assert np.array_equal(got[:, :, 0], ub[:, j]) and np.array_equal(got[:, :, 1], ub[:, j])
    assert np.array_equal(got[:, :, 2], yb[:, 2 * j]) and np.array_equal(got[:, :, 6], yb[:, 2 * j + 1])
    assert np.array_equal(got[:, :, 4], vb[:, j])
    # crop and edge extension
    big = O.pack_wide(pl, 0, 0, w + 5, h + 3, 0x103).reshape(h + 3, w + 5, 4)
    assert np.array_equal(big[:h, :w], O.pack_wide(pl, 0, 0, w, h, 0x103).reshape(h, w, 4))
    assert np.array_equal(big[h:, :w], np.repeat(big[h - 1:h, :w], 3, axis=0))
    assert np.array_equal(big[:, w:], np.repeat(big[:, w - 1:w], 5, axis=1))
