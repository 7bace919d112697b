"""CPU: the low-delay oracle (oracle/oracle_lowdelay.c) against what pins it.

  * quantiser tables: all 61 entries of schro_table_quant / schro_table_offset_1_2
    (tests/golden/quant_tables.json holds the reference's numbers);
  * the fast decoder's 16-bit dequantisation: the reference's own
    orc_dequantise_var_s16_ip, compiled unmodified into oracle/_ref;
  * exp-Golomb codes: the code table of the Dirac specification (A.4.3: 0 -> 1, 1 -> 001,
    2 -> 011, 3 -> 00001 ..., then the sign bit), hand-packed into a slice;
  * composition (slice offsets, sub-band / slice rectangles, U/V interleave, DC prediction):
    an independent numpy model of the same syntax.
The slice syntax as a whole has no reference-made vector (parity unpinned).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_quant_tables_match_the_reference_constants():
    want = json.load(open(os.path.join(HERE, "golden", "quant_tables.json")))
    factor, offset = O.quant_tables()
    assert factor == want["schro_table_quant"]
    assert offset == want["schro_table_offset_1_2"]


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built")
def test_fast_dequantisation_matches_compiled_reference_kernel():
    ref = O.reforc().orc_dequantise_var_s16_ip
    ref.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    factor, offset = O.quant_tables()
    q = np.concatenate([np.arange(-300, 301), synth.full_range(1, 4000, np.int16, seed=9).ravel(),
                        np.array([-32768, 32767, -32767, 16384, -16384])]).astype(np.int16)
    for qi in range(61):
        got = q.copy()
        # as schro_lowdelay_init_quant_arrays stores them: int16_t, offset + 2 (schrolowdelay.c:478-479)
        f = np.full(q.size, factor[qi] & 0xffff, np.uint16).view(np.int16)
        o = np.full(q.size, (offset[qi] + 2) & 0xffff, np.uint16).view(np.int16)
        ref(got.ctypes.data, f.ctypes.data, o.ctypes.data, q.size)
        mine = np.array([O.lib().oracle_dequantise_var_s16(int(v), factor[qi], offset[qi]) for v in q], np.int16)
        assert np.array_equal(got, mine), qi


def pack_bits(bits, nbytes, pad="1"):
    bits = bits + pad * (8 * nbytes - len(bits))
    assert len(bits) == 8 * nbytes
    return np.array([int(bits[k:k + 8], 2) for k in range(0, len(bits), 8)], np.uint8)


# unsigned interleaved exp-Golomb codes of the specification's table
SPEC_CODES = {0: "1", 1: "001", 2: "011", 3: "00001", 4: "00011", 5: "01001", 6: "01011", 7: "0000001",
              8: "0000011", 9: "0001001"}


def test_spec_code_table_in_a_hand_packed_slice():
    # one slice, depth 0: 4x2 luma, 2x2 chroma (4:2:0), base index 0 -> factor 4, offset 1:
    # (4 |q| + 1 + 2) >> 2 = |q|
    P = dict(transform_depth=0, iwt_luma_width=4, iwt_luma_height=2, iwt_chroma_width=2, iwt_chroma_height=1,
             n_horiz_slices=1, n_vert_slices=1, slice_bytes_num=16, slice_bytes_denom=1, quant_matrix=[0])
    luma = [0, 1, -2, 3, -4, 5, 9, -7]
    chroma = [6, -8, 0, -1]                    # U0 V0 U1 V1

    def code(v):
        return SPEC_CODES[abs(v)] + ("" if v == 0 else ("1" if v < 0 else "0"))
    ybits = "".join(code(v) for v in luma)
    length_bits = 8                            # ilog2up (8 * 16 = 128) = 8
    bits = format(0, "07b") + format(len(ybits), "0%db" % length_bits) + ybits + "".join(code(v) for v in chroma)
    data = pack_bits(bits, 16)
    planes = [np.zeros((2, 4), np.int16), np.zeros((1, 2), np.int16), np.zeros((1, 2), np.int16)]
    O.lowdelay_decode(data, planes, P)
    # depth 0: the whole plane is the LL band -> undo nothing, predict forward instead
    want = [O.dc_predict(np.array(luma, np.int16).reshape(2, 4)),
            O.dc_predict(np.array(chroma[0::2], np.int16).reshape(1, 2)),
            O.dc_predict(np.array(chroma[1::2], np.int16).reshape(1, 2))]
    for got, w in zip(planes, want):
        assert np.array_equal(got, w)
    # row 0 is a running sum, row 1 uses the mean of three neighbours
    assert planes[0][0].tolist() == [0, 1, -1, 2]


def dc_predict_model(a):
    """Per-sample Python loop (schrodecoder.c:3219-3277) -- small bands only."""
    h, w = a.shape
    wrap = (lambda v: ((int(v) + 32768) & 0xffff) - 32768) if a.dtype == np.int16 else \
        (lambda v: ((int(v) + (1 << 31)) & 0xffffffff) - (1 << 31))
    x = [[int(v) for v in row] for row in a]
    for j in range(h):
        for i in range(w):
            if j == 0:
                pred = x[0][i - 1] if i else 0
            elif i == 0:
                pred = x[j - 1][0]
            else:
                s = x[j][i - 1] + x[j - 1][i] + x[j - 1][i - 1] + 1
                pred = (s * 21845 + 10922) >> 16 if a.dtype == np.int16 else s // 3
            x[j][i] = wrap(x[j][i] + pred)
    return np.array(x, a.dtype)


@pytest.mark.parametrize("dtype", [np.int16, np.int32])
def test_dc_predict_against_loop_model(dtype):
    for (h, w, seed) in [(1, 1, 1), (1, 9, 2), (7, 1, 3), (5, 8, 4), (17, 23, 5)]:
        a = synth.image_s(h, w, dtype, seed=seed) * 3
        assert np.array_equal(O.dc_predict(a), dc_predict_model(a))
    # s16 sums wrap like the reference's int16_t stores
    a = synth.full_range(6, 7, dtype, seed=8)
    if dtype == np.int32:
        a = a >> 6          # keep every sum inside int (the reference adds ints)
    assert np.array_equal(O.dc_predict(a), dc_predict_model(a))


def subband_view(plane, depth, index):
    """schro_subband_get_frame_data on a numpy plane."""
    position = [0, 1, 2, 3, 5, 6, 7, 9, 10, 11, 13, 14, 15, 17, 18, 19, 21, 22, 23, 25, 26, 27][index]
    shift = depth - (position >> 2)
    h, w = plane.shape[0] >> shift, plane.shape[1] >> shift
    rows = plane[(1 << shift) >> 1 if position & 2 else 0::1 << shift]
    return rows[:h, w if position & 1 else 0:][:, :w]


def numpy_model(q, P, bpp, base_index, arith):
    """Dequantised + DC-predicted planes for slices that hold all their codes."""
    factor, offset = O.quant_tables()
    dt = np.int16 if bpp == 2 else np.int32
    out = [np.zeros(p.shape, dt) for p in q]
    nh, nv, depth = P["n_horiz_slices"], P["n_vert_slices"], P["transform_depth"]
    for k in range(3):
        for i in range(1 + 3 * depth):
            src, dst = subband_view(q[k], depth, i), subband_view(out[k], depth, i)
            h, w = src.shape
            for sy in range(nv):
                for sx in range(nh):
                    ys, xs = slice(h * sy // nv, h * (sy + 1) // nv), slice(w * sx // nh, w * (sx + 1) // nh)
                    qi = min(max(int(base_index[sy * nh + sx]) - P["quant_matrix"][i], 0), 60)
                    v = src[ys, xs].astype(np.int64)
                    if arith == O.LOWDELAY_FAST16:
                        v16 = v.astype(np.int16).astype(np.int64)
                        t = (np.abs(v16) * np.int64(np.int16(np.uint16(factor[qi] & 0xffff)))).astype(np.int16)
                        t = (t.astype(np.int64) + np.int64(np.int16(np.uint16((offset[qi] + 2) & 0xffff)))).astype(np.int16)
                        d = ((t >> 2).astype(np.int64) * np.sign(v16)).astype(np.int16)
                    else:
                        # int arithmetic: the sum wraps at 32 bits, the shift is arithmetic
                        t = ((np.abs(v) * factor[qi] + offset[qi] + 2) & 0xffffffff).astype(np.uint32).view(np.int32)
                        d = np.sign(v) * (t >> 2).astype(np.int64)
                    dst[ys, xs] = d.astype(dt)
        ll = subband_view(out[k], depth, 0)
        ll[...] = O.dc_predict(np.ascontiguousarray(ll))
    return out


CASES = [
    # w, h, chroma, depth, slice w, slice h, bytes num, denom, bpp, slices override
    (64, 32, (1, 1), 2, 16, 8, 241, 2, 2, None),         # fast16, fractional slice size
    (64, 32, (0, 0), 1, 8, 8, 200, 1, 2, None),          # fast16 4:4:4
    (72, 40, (1, 0), 2, 24, 10, 1000, 3, 2, (5, 3)),     # slow16: ragged slice rectangles
    (64, 48, (1, 0), 3, 16, 16, 500, 1, 4, None),        # s32
    (96, 64, (1, 1), 4, 32, 32, 2001, 4, 4, (5, 3)),     # s32 ragged, depth 4
    (16, 16, (1, 1), 0, 8, 8, 150, 1, 2, None),          # depth 0: one band
]


@pytest.mark.parametrize("case", CASES)
def test_decode_against_numpy_model(case):
    w, h, chroma, depth, sw, sh, num, den, bpp, override = case
    P = synth.lowdelay_params(w, h, chroma, depth, sw, sh, num, den)
    if override:
        P["n_horiz_slices"], P["n_vert_slices"] = override
    arith = O.lowdelay_arith(P, bpp)
    assert arith == (O.LOWDELAY_S32 if bpp == 4 else O.LOWDELAY_SLOW16 if override else O.LOWDELAY_FAST16)
    q = synth.quantised_planes(P, seed=w + depth, scale=0.8, big_every=61, big_range=1 << (13 if bpp == 2 else 20))
    bi = synth.lowdelay_base_index(P, seed=h, lo=0, hi=48)
    data = O.lowdelay_write(q, P, bpp, bi)
    assert data.size == O.lowdelay_slice_bytes(P)
    got = [np.full(p.shape, 0x55, np.int16 if bpp == 2 else np.int32) for p in q]
    O.lowdelay_decode(data, got, P)
    want = numpy_model(q, P, bpp, bi, arith)
    for k in range(3):
        assert np.array_equal(got[k], want[k]), "component %d" % k


def test_guard_bits_and_short_slices():
    # slices too small for their codes: what is cut off reads as guard bits -> zeros, and a
    # code cut in the middle continues with ones (schrounpack.c:80-88)
    P = synth.lowdelay_params(32, 16, (1, 1), 1, 16, 8, 12)
    q = synth.quantised_planes(P, seed=3, scale=2.0)
    bi = synth.lowdelay_base_index(P, seed=3, lo=0, hi=0)
    for pad in (0, 1):
        data = O.lowdelay_write(q, P, 2, bi, pad_bit=pad)
        got = [np.zeros(p.shape, np.int16) for p in q]
        O.lowdelay_decode(data, got, P)
        # chroma never made it into a 12-byte slice: every sample 0 before DC prediction
        assert not got[1].any() and not got[2].any()
        assert got[0].any()


def test_golden_fixture_is_what_the_oracle_decodes():
    z = np.load(os.path.join(HERE, "golden", "lowdelay_oracle.npz"))
    names = ("transform_depth", "iwt_luma_width", "iwt_luma_height", "iwt_chroma_width", "iwt_chroma_height",
             "n_horiz_slices", "n_vert_slices", "slice_bytes_num", "slice_bytes_denom")
    for case, bpp in (("fast16", 2), ("slow16", 2), ("s32", 4)):
        v = z[case + "_params"].tolist()
        P = dict(zip(names, v[:9]), quant_matrix=v[9:])
        planes = [np.zeros_like(z["%s_comp%d" % (case, k)]) for k in range(3)]
        O.lowdelay_decode(z[case + "_slices"], planes, P)
        assert O.lowdelay_arith(P, bpp) == {"fast16": 0, "slow16": 1, "s32": 2}[case]
        for k in range(3):
            assert np.array_equal(planes[k], z["%s_comp%d" % (case, k)])


def test_slice_codes_read_by_the_stream_validated_reader():
    # oracle/dirac_stream.py's Bits is the reader that parses the reference's test stream (whose
    # decoded frames carry the reference decoder's digests): the same interleaved exp-Golomb
    # reader, restated independently of oracle_lowdelay.c.  The codes of a written slice, read
    # with it, must be the quantised values that went in -- header fields included.
    import sys
    sys.path.insert(0, os.path.join(O.ROOT, "oracle"))
    import dirac_stream as D
    P = synth.lowdelay_params(32, 16, (1, 1), 1, 16, 8, 400)
    q = synth.quantised_planes(P, seed=21, scale=2.0, big_every=7, big_range=1 << 20)
    bi = synth.lowdelay_base_index(P, seed=5, lo=0, hi=127)
    data = O.lowdelay_write(q, P, 2, bi)
    nh, nv, depth = P["n_horiz_slices"], P["n_vert_slices"], P["transform_depth"]
    for s in range(nh * nv):
        sy, sx = divmod(s, nh)
        b = D.Bits(bytes(data[400 * s:400 * (s + 1)]))
        assert b.bits(7) == bi[s]
        ylen = b.bits(12)                               # ilog2up (8 * 400) = 12
        start = b.p
        for k in range(2):
            for i in range(1 + 3 * depth):
                bands = [subband_view(q[k + c], depth, i) for c in range(k + 1)]
                h, w = bands[0].shape
                for y in range(h * sy // nv, h * (sy + 1) // nv):
                    for x in range(w * sx // nh, w * (sx + 1) // nh):
                        for band in bands:
                            assert b.sint() == band[y, x]
            if k == 0:
                assert b.p - start == ylen


def test_synth_slice_writer_against_the_oracle_decoder():
    """bench.py's lowdelay_8k key makes its slices with tests/synth.py's writer (no oracle/ in bench.py's
    product legs): what it writes must be what the oracle's decoder -- the restated reference reader --
    reads, for every sub-band but the DC-predicted LL band, and the LL band before prediction."""
    import json
    P = synth.lowdelay_params(256, 64, (1, 0), 3, 32, 8, 155, 1)
    data, kind, made = synth.lowdelay_picture(P, seed=11, kinds=5)
    tables = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "quant_tables.json")))
    dims = [(P["iwt_luma_height"], P["iwt_luma_width"])] + [(P["iwt_chroma_height"], P["iwt_chroma_width"])] * 2
    got = [np.zeros(d, np.int32) for d in dims]
    O.lowdelay_decode(data, got, P)
    nx, ny, depth = P["n_horiz_slices"], P["n_vert_slices"], P["transform_depth"]
    for sy in range(ny):
        for sx in range(nx):
            base, vals = made[int(kind[sy, sx])]
            for comp, (h, w) in enumerate(dims):
                for index in range(1, 1 + 3 * depth):
                    _, c0, r0, step, bw, bh = synth.subband_geometry(w, h, depth, index)
                    x0, x1, y0, y1 = bw * sx // nx, bw * (sx + 1) // nx, bh * sy // ny, bh * (sy + 1) // ny
                    want = synth.lowdelay_expected_band(P, comp, index, base, vals, tables)
                    assert np.array_equal(got[comp][r0 + step * y0:r0 + step * y1:step, c0 + x0:c0 + x1], want), (sy, sx, comp, index)
