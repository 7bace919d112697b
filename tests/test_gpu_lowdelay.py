"""GPU parity: VC-2 low-delay transform data (slice unpack + dequantise + DC prediction,
schro_hip_lowdelay_batch through the C ABI) vs the CPU oracle's restatement of
schro_decoder_decode_lowdelay_transform_data (schrolowdelay.c:559-762).  Bit-exact, for
each of the reference's three slice decoders (s16 fast / s16 slow / s32)."""
import os

import numpy as np
import pytest

import oracle_lib as O
import schroedinger_amd as sa
import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


class Offset:
    """A byte range inside a device buffer (slices need not start 4-byte aligned)."""

    def __init__(self, plane, off, nbytes):
        self.ptr, self.width = plane.ptr + off, nbytes


def decode_gpu(ctx, pictures, P, bpp, misalign=0):
    """pictures: slice byte arrays.  Returns per picture the three decoded planes."""
    dt = np.int16 if bpp == 2 else np.int32
    jobs, keep = [], []
    for n, data in enumerate(pictures):
        off = (misalign + n) % 4 if misalign else 0
        raw = np.concatenate([np.full(off, 0xa5, np.uint8), data])
        d = ctx.upload_bytes(raw)
        planes = [ctx.plane(P["iwt_chroma_height"] if k else P["iwt_luma_height"],
                            P["iwt_chroma_width"] if k else P["iwt_luma_width"], dt).fill(0x5a) for k in range(3)]
        jobs.append((Offset(d, off, data.size), planes))
        keep.append(d)
    ctx.lowdelay_batch(jobs, P)
    out = [[p.download() for p in planes] for _, planes in jobs]
    for (_, planes), d in zip(jobs, keep):
        d.free()
        for p in planes:
            p.free()
    return out


def decode_cpu(data, P, bpp):
    planes = [np.full((P["iwt_chroma_height"] if k else P["iwt_luma_height"],
                       P["iwt_chroma_width"] if k else P["iwt_luma_width"]), 0x5a5a5a5a & (0xffff if bpp == 2 else -1),
                      np.int16 if bpp == 2 else np.int32) for k in range(3)]
    O.lowdelay_decode(data, planes, P)
    return planes


def compare(got, want, what):
    for k in range(3):
        if not np.array_equal(got[k], want[k]):
            bad = np.argwhere(got[k] != want[k])
            raise AssertionError("%s component %d: %d mismatches, first at (y,x)=%s got %d want %d" % (
                what, k, len(bad), tuple(bad[0]), got[k][tuple(bad[0])], want[k][tuple(bad[0])]))


# w, h, chroma, depth, slice w, slice h, bytes num, denom, slices override
GEOMETRIES = [
    (64, 32, (1, 1), 2, 16, 8, 241, 2, None),
    (64, 32, (0, 0), 1, 8, 8, 200, 1, None),
    (128, 64, (1, 0), 3, 32, 8, 321, 3, None),
    (72, 40, (1, 0), 2, 24, 10, 1000, 3, (5, 3)),       # ragged slice rectangles (slow decoder for s16)
    (96, 64, (1, 1), 4, 32, 32, 2001, 4, (5, 3)),
    (16, 16, (1, 1), 0, 8, 8, 150, 1, None),            # depth 0
    (200, 24, (1, 1), 1, 8, 8, 90, 1, (25, 3)),         # more slices per row than a wave has lanes... of 3 rows
    (1040, 16, (1, 0), 1, 16, 16, 255, 2, None),        # 65 slices in a row: two waves, field width changes (255/2)
    (256, 32, (1, 0), 2, 64, 8, 300, 1, None),          # rectangle rows of 32 values: 32 staging words per lane
    (512, 32, (1, 0), 1, 128, 8, 1201, 2, None),        # rows of 64 values: 64 words
    (512, 16, (0, 0), 1, 128, 8, 900, 1, None),         # U and V rows of 128 values together: slice_kernel takes it
]


@pytest.fixture(params=["runs", "values"])
def slice_kernel(request, monkeypatch):
    """Both slice kernels: slice_run_kernel (a step per non-zero value) where the geometry allows it,
    and slice_kernel (a step per value) for everything."""
    if request.param == "values":
        monkeypatch.setenv("SCHRO_HIP_SLICE_RUNS", "0")
    return request.param


@pytest.mark.parametrize("bpp", [2, 4])
@pytest.mark.parametrize("geo", GEOMETRIES)
def test_legal_streams(ctx, geo, bpp, slice_kernel):
    w, h, chroma, depth, sw, sh, num, den, override = geo
    P = synth.lowdelay_params(w, h, chroma, depth, sw, sh, num, den)
    if override:
        P["n_horiz_slices"], P["n_vert_slices"] = override
    assert ctx.lowdelay_arith(P, bpp) == O.lowdelay_arith(P, bpp)
    pictures = []
    for n in range(3):
        q = synth.quantised_planes(P, seed=w + depth + 11 * n, scale=0.7 + 0.4 * n)
        bi = synth.lowdelay_base_index(P, seed=h + n, lo=0, hi=44)
        pictures.append(O.lowdelay_write(q, P, bpp, bi, pad_bit=n & 1))
    got = decode_gpu(ctx, pictures, P, bpp, misalign=1)
    for n, data in enumerate(pictures):
        compare(got[n], decode_cpu(data, P, bpp), "picture %d" % n)


@pytest.mark.parametrize("bpp", [2, 4])
def test_seeded_geometries(ctx, bpp, slice_kernel):
    # two dozen seeded pictures: depth 1 .. 4, slices of 8 .. 128 x 4 .. 32 luma samples that divide every
    # sub-band evenly (slice_run_kernel's case: 16, 32 or 64 staging words per lane, or -- U and V rows
    # beyond 64 words -- slice_kernel), 4:2:0 / 4:2:2 / 4:4:4, byte budgets from starved (codes cut off
    # by the slice's end) to roomy, sparse to dense values with the odd long code, slice_y_length
    # fields that lie now and then
    rng = np.random.default_rng(1234 + bpp)
    for case in range(24):
        depth = int(rng.integers(1, 5))
        sw = int(rng.choice([8, 16, 32, 64, 128])) * (1 if depth < 4 else 2)
        sh = int(rng.choice([4, 8, 16, 32]))
        sw, sh = max(sw, 1 << depth), max(sh, 1 << depth)
        chroma = [(1, 1), (1, 0), (0, 0)][int(rng.integers(0, 3))]
        if (sw >> chroma[0]) < (1 << depth) or (sh >> chroma[1]) < (1 << depth):
            chroma = (0, 0)
        nx, ny = int(rng.integers(1, 70 if sw <= 16 else 9)), int(rng.integers(1, 5))
        w, h = sw * nx, sh * ny
        num = int(rng.integers(max(8, sw * sh // 24), max(12, sw * sh // 2)))
        den = int(rng.choice([1, 1, 2, 3]))
        P = synth.lowdelay_params(w, h, chroma, depth, sw, sh, num, den)
        assert P["n_horiz_slices"] == nx and P["n_vert_slices"] == ny
        q = synth.quantised_planes(P, seed=case, scale=float(rng.choice([0.3, 0.8, 2.0, 6.0])),
                                   big_every=int(rng.choice([0, 0, 97, 1031])), big_range=1 << int(rng.integers(17, 31)))
        bi = synth.lowdelay_base_index(P, seed=case, lo=0, hi=int(rng.choice([20, 60, 127])))
        data = O.lowdelay_write(q, P, bpp, bi, pad_bit=case & 1, y_length_bias=int(rng.choice([0, 0, 0, -9, 31])))
        got = decode_gpu(ctx, [data], P, bpp, misalign=case % 3)
        compare(got[0], decode_cpu(data, P, bpp), "case %d: %dx%d depth %d slices %dx%d chroma %s %d/%d bytes" % (
            case, w, h, depth, sw, sh, chroma, num, den))


@pytest.mark.parametrize("bpp", [2, 4])
def test_long_codes_wrap_and_quantiser_range(ctx, bpp, slice_kernel):
    # values beyond the 32-bit decode window (|v| >= 65535), products that wrap in 16 / 32 bits,
    # base indices up to 127 (quantiser index clamps at 60; schrolowdelay.c:140)
    for override in (None, (5, 3)):
        P = synth.lowdelay_params(96, 48, (1, 0), 2, 24, 12, 3000, 1)
        if override:
            P["n_horiz_slices"], P["n_vert_slices"] = override
        q = synth.quantised_planes(P, seed=5, scale=1.5, big_every=13, big_range=1 << 31)
        q[0][3, 5], q[0][7, 9], q[1][2, 2] = (1 << 31) - 1, -(1 << 31) + 1, 65535
        bi = synth.lowdelay_base_index(P, seed=6, lo=0, hi=127)
        data = O.lowdelay_write(q, P, bpp, bi)
        got = decode_gpu(ctx, [data], P, bpp)
        compare(got[0], decode_cpu(data, P, bpp), "override %s" % (override,))


@pytest.mark.parametrize("bpp", [2, 4])
def test_short_slices_and_corrupt_lengths(ctx, bpp, slice_kernel):
    # codes cut off by the end of the slice (guard bits), slice_y_length fields that point
    # short of / beyond the luma codes and beyond the slice (the reference then reads the next
    # slice's bytes; at the end of the buffer it would read out of bounds, we read guard bits)
    P = synth.lowdelay_params(128, 32, (1, 1), 2, 16, 8, 49, 2)
    q = synth.quantised_planes(P, seed=8, scale=2.5)
    bi = synth.lowdelay_base_index(P, seed=9, lo=0, hi=30)
    pictures = [O.lowdelay_write(q, P, bpp, bi, pad_bit=pad, y_length_bias=bias)
                for pad in (0, 1) for bias in (0, -17, 23, 150, 4000)]
    rnd = (synth.lcg(pictures[0].size, 4) & 0xff).astype(np.uint8)       # noise is a stream too
    pictures += [rnd, np.zeros_like(rnd), np.full_like(rnd, 0xff)]
    got = decode_gpu(ctx, pictures, P, bpp, misalign=2)
    for n, data in enumerate(pictures):
        compare(got[n], decode_cpu(data, P, bpp), "stream %d" % n)


def test_golden_fixture(ctx):
    z = np.load(os.path.join(HERE, "golden", "lowdelay_oracle.npz"))
    names = ("transform_depth", "iwt_luma_width", "iwt_luma_height", "iwt_chroma_width", "iwt_chroma_height",
             "n_horiz_slices", "n_vert_slices", "slice_bytes_num", "slice_bytes_denom")
    for case, bpp in (("fast16", 2), ("slow16", 2), ("s32", 4)):
        v = z[case + "_params"].tolist()
        P = dict(zip(names, v[:9]), quant_matrix=v[9:])
        got = decode_gpu(ctx, [z[case + "_slices"]], P, bpp)[0]
        compare(got, [z["%s_comp%d" % (case, k)] for k in range(3)], case)


@pytest.mark.parametrize("dtype", [np.int16, np.int32])
def test_dc_predict_alone(ctx, dtype):
    shapes = [(1, 1), (1, 300), (300, 1), (64, 64), (65, 130), (540, 960), (1100, 37), (2100, 3)]
    planes, want = [], []
    for n, (h, w) in enumerate(shapes):
        a = synth.image_s(h, w, dtype, seed=20 + n) * 5
        if n % 2 and dtype == np.int16:
            a = synth.full_range(h, w, dtype, seed=40 + n)           # int16 sums wrap
        planes.append(ctx.upload(a))
        want.append(O.dc_predict(a))
    ctx.dc_predict_batch(planes)
    for p, wnt, shp in zip(planes, want, shapes):
        got = p.download()
        assert np.array_equal(got, wnt), shp
        p.free()


@pytest.mark.parametrize("kernel", ["skew", "barrier"])
@pytest.mark.parametrize("dtype", [np.int16, np.int32])
def test_dc_predict_strips(ctx, dtype, kernel, monkeypatch):
    # bands of whole 16-byte pieces: dc_skew_kernel (strips of 64 rows on separate CUs, the last row
    # handed from strip to strip) -- one strip, many, a last strip of one row, rows shorter than a block,
    # rows longer than the rings, a last block that is not whole; full-range values (every sum wraps);
    # twice, so that the second launch meets the first one's hand-over buffer.  "barrier": the same
    # through dc_predict_kernel
    if kernel == "barrier":
        monkeypatch.setenv("SCHRO_HIP_DC_SKEW", "0")
    E = 16 // np.dtype(dtype).itemsize
    shapes = [(1, E), (3, 2 * E), (64, 64), (65, 136), (129, 5 * E), (540, 960), (1100, 40), (200, 2048), (70, 16 * 40 + E),
              # the in-ring's wrap (16 blocks of 16 samples) and its neighbours; strips of 63 / 64 / 127 / 128 rows
              (63, 256), (64, 240), (127, 272), (128, 16 * 33), (66, 16 * 32 - E)]
    for rnd in range(2):
        planes, want = [], []
        for n, (h, w) in enumerate(shapes):
            a = synth.full_range(h, w, dtype, seed=50 + n + 100 * rnd) if n % 2 else synth.image_s(h, w, dtype, seed=20 + n) * 5
            planes.append(ctx.upload(a))
            want.append(O.dc_predict(a))
        ctx.dc_predict_batch(planes)
        for p, wnt, shp in zip(planes, want, shapes):
            got = p.download()
            assert np.array_equal(got, wnt), (shp, rnd)
            p.free()


def test_intra_picture_through_the_frame_layer(ctx):
    # a low-delay picture the way a patched schrodecoder.c would run it: compressed slices ->
    # transform frame on the device (schro_decoder_decode_lowdelay_transform_data), inverse
    # wavelet in place of x_wavelet_transform, u8 picture (schrodecoder.c:1788-1790)
    import ctypes as C
    from schroedinger_amd import frames
    w, h, depth, filt = 160, 96, 3, 1
    P = synth.lowdelay_params(w, h, (1, 0), depth, 16, 16, 400)
    for bpp, fmt in ((2, sa.FORMAT_S16_422), (4, sa.FORMAT_S32_422)):
        dt = np.int16 if bpp == 2 else np.int32
        q = synth.quantised_planes(P, seed=12 + bpp, scale=1.1)
        data = O.lowdelay_write(q, P, bpp, synth.lowdelay_base_index(P, seed=3, lo=0, hi=20))
        tf = frames.DeviceFrame(ctx, fmt, P["iwt_luma_width"], P["iwt_luma_height"])
        sa.check(ctx.lib.schro_hip_decode_lowdelay_transform_data(
            tf.ptr(), data.ctypes.data_as(C.c_void_p), data.size, C.byref(ctx.lowdelay_params(P))))
        coeffs = decode_cpu(data, P, bpp)
        got = tf.download()
        for k in range(3):
            assert np.array_equal(got[k], coeffs[k]), (bpp, k)
        params = frames.make_params(wavelet_filter_index=filt, transform_depth=depth,
                                    iwt_luma_width=P["iwt_luma_width"], iwt_luma_height=P["iwt_luma_height"],
                                    iwt_chroma_width=P["iwt_chroma_width"], iwt_chroma_height=P["iwt_chroma_height"])
        frame = frames.DeviceFrame(ctx, fmt, P["iwt_luma_width"], P["iwt_luma_height"])
        sa.check(ctx.lib.schro_frame_inverse_iwt_transform_hip(frame.ptr(), tf.ptr(), C.byref(params)))
        pix = frame.download()
        for k in range(3):
            assert np.array_equal(pix[k], O.inverse_iwt(coeffs[k].astype(dt), depth, filt)), (bpp, k)
        frame.unref()
        tf.unref()


def test_bad_arguments_are_refused(ctx):
    P = synth.lowdelay_params(64, 32, (1, 1), 2, 16, 8, 100)
    data = np.zeros(O.lowdelay_slice_bytes(P), np.uint8)
    with pytest.raises(sa.SchroHipError):                    # buffer shorter than the slices
        decode_gpu(ctx, [data[:-1]], P, 2)
    bad = dict(P, iwt_luma_width=66)
    with pytest.raises(sa.SchroHipError):                    # not a multiple of 2^depth
        decode_gpu(ctx, [data], bad, 2)
    bad = dict(P, slice_bytes_denom=0)
    with pytest.raises(sa.SchroHipError):
        decode_gpu(ctx, [data], bad, 2)


def test_8k_422_10bit_configuration(ctx):
    # BASELINE config 5: 7680x4320 4:2:2, s32 coefficients, 3 levels, 32x8-sample slices
    # (240 x 540 = 129600 slices), about 2.4 bits per sample
    P = synth.lowdelay_params(7680, 4320, (1, 0), 3, 32, 8, 155, 1)
    q = synth.quantised_planes(P, seed=31, scale=0.9)
    bi = synth.lowdelay_base_index(P, seed=32, lo=4, hi=28)
    data = O.lowdelay_write(q, P, 4, bi)
    del q
    got = decode_gpu(ctx, [data], P, 4)
    compare(got[0], decode_cpu(data, P, 4), "8K")


def test_config5_whole_pixel_path_8k(ctx):
    # BASELINE config 5 end to end on the device, against the oracle's chain on the host:
    # slices -> coefficients (+ DC prediction) -> 3-level Haar (no shift) s32 inverse wavelet
    # -> v210 copy-out, at 7680x4320 4:2:2 (what tests/bench_lowdelay.py times).
    W, H, depth, filt = 7680, 4320, 3, 3
    P = synth.lowdelay_params(W, H, (1, 0), depth, 32, 8, 155, 1)
    q = synth.quantised_planes(P, seed=41, scale=0.9)
    bi = synth.lowdelay_base_index(P, seed=42, lo=4, hi=28)
    data = O.lowdelay_write(q, P, 4, bi)
    del q
    dims = [(P["iwt_luma_height"], P["iwt_luma_width"])] + [(P["iwt_chroma_height"], P["iwt_chroma_width"])] * 2
    d = ctx.upload_bytes(data)
    co = [ctx.plane(h, w, np.int32).fill(0x5a) for (h, w) in dims]
    px = [ctx.plane(h, w, np.int32) for (h, w) in dims]
    v210 = ctx.plane(H, 16 * (-(-W // 6)), np.uint8).fill(0x5a)
    ctx.lowdelay_batch([(d, co)], P)
    ctx.iiwt_batch(list(zip(co, px)), depth, filt)
    ctx.pack_v210_batch([(px, 1, 0, v210, W, H)])
    got_px = [p.download() for p in px]
    got_v210 = v210.download()
    # r05 -- the fused route: the three Haar levels with the copy-out as their epilogue, no pixel frame
    fused = ctx.plane(H, 16 * (-(-W // 6)), np.uint8).fill(0xa5)
    ctx.iiwt_pack_v210_batch([(co, 1, 0, fused, W, H)], depth, filt)
    got_fused = fused.download()
    for p in [d, v210, fused] + co + px:
        p.free()
    want_co = decode_cpu(data, P, 4)
    want_px = [O.inverse_iwt(c, depth, filt) for c in want_co]
    for k in range(3):
        assert np.array_equal(got_px[k], want_px[k]), "component %d after the inverse wavelet" % k
    want_v210 = O.pack_v210(want_px, 1, 0, W, H)
    assert np.array_equal(got_v210, want_v210), "two passes"
    assert np.array_equal(got_fused, want_v210), "fused"


@pytest.mark.parametrize("filt", [3, 4])
def test_transform_and_v210_copy_out_in_one_call(ctx, filt):
    """r05: schro_hip_iiwt_pack_v210_batch == schro_hip_iiwt_batch + schro_hip_pack_v210_batch == the oracle's chain, on the
    fused kernel (s32 Haar, depth 3, 4:2:2, sizes that are multiples of 48 x 8: one strip piece, several, a partial last one;
    full-range values: the s32 -> s16 truncation and the 10-bit clamp) and on the two-pass fallback (other sizes, depths,
    filters, sample sizes, chroma formats, pictures smaller than the transform)."""
    def run(w, h, depth, f, dtype, hs, vs, ow=None, oh=None, full=False, batch=1):
        ow, oh = ow or w, oh or h
        dims = [(h, w), (h >> vs, w >> hs), (h >> vs, w >> hs)]
        jobs, wants, keep = [], [], []
        for n in range(batch):
            if full:
                co = [synth.lcg(a * b, 7 + 3 * n + k).astype(np.int64).reshape(a, b) for k, (a, b) in enumerate(dims)]
                co = [((c * 2654435761) % (1 << 20) - (1 << 19)).astype(dtype) for c in co]
            else:
                co = [O.forward_iwt((synth.image_s(a, b, dtype, seed=5 + 3 * n + k).astype(np.int64) * 3).astype(dtype), depth, f)
                      for k, (a, b) in enumerate(dims)]
            d_co = [ctx.upload(c) for c in co]
            dst = ctx.plane(oh, 16 * (-(-ow // 6)), np.uint8).fill(0x3c)
            jobs.append((d_co, hs, vs, dst, ow, oh))
            px = [O.inverse_iwt(c, depth, f) for c in co]
            wants.append(O.pack_v210([p[:(oh if k == 0 else -(-oh >> vs)), :(ow if k == 0 else -(-ow >> hs))] for k, p in enumerate(px)],
                                     hs, vs, ow, oh))
            keep += d_co + [dst]
        ctx.iiwt_pack_v210_batch(jobs, depth, f)
        for n, (j, want) in enumerate(zip(jobs, wants)):
            assert np.array_equal(j[3].download(), want), (w, h, depth, f, np.dtype(dtype).name, hs, vs, ow, oh, n)
        [p.free() for p in keep]
    # the fused kernel
    run(48, 8, 3, filt, np.int32, 1, 0)
    run(960, 64, 3, filt, np.int32, 1, 0)
    run(1920, 1080 // 8 * 8, 3, filt, np.int32, 1, 0, batch=2)         # two strip pieces per row, the second one partial... (1920 = 2 x 960)
    run(1104, 72, 3, filt, np.int32, 1, 0, full=True)                   # 960 + 144: a partial piece; values beyond 16 bits
    run(2064, 40, 3, filt, np.int32, 1, 0, batch=3)
    # the two passes: not depth 3, s16, a width that is no multiple of 48, a picture inside the transform, another filter
    run(96, 32, 2, filt, np.int32, 1, 0)
    run(96, 32, 3, filt, np.int16, 1, 0)
    run(64, 32, 3, filt, np.int32, 1, 0)
    co = [ctx.plane(32, 96, np.int32), ctx.plane(16, 48, np.int32), ctx.plane(16, 48, np.int32)]
    dst = ctx.plane(32, 16 * 16, np.uint8)
    with pytest.raises(sa.SchroHipError, match="4:2:2"):
        ctx.iiwt_pack_v210_batch([(co, 1, 1, dst, 96, 32)], 3, filt)
    [p.free() for p in co + [dst]]
    run(96, 32, 3, filt, np.int32, 1, 0, ow=90, oh=30)
    run(96, 32, 3, 1, np.int32, 1, 0)
