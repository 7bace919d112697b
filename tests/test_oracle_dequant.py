"""CPU: the oracle's core-syntax dequantisation (oracle/oracle_dequant.c).

Pins: the reference's COMPILED kernels orc_dequantise_s16_2d_8xn / _4xn / _s16_ip_2d /
_s32_ip_2d (oracle/_ref: schroorc-dist.c built unmodified); schro_table_offset_3_8 against the
reference's numbers; and the reference's own stream: its quantised values, dequantised (and, for
intra pictures, DC-predicted) by the oracle, are the coefficients whose decoded pictures carry
the reference decoder's MD5s (tests/test_oracle_stream.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import stream_lib as S

needs_ref = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


def test_offset_3_8_table_is_the_references():
    want = json.load(open(os.path.join(S.GOLDEN, "arith_lut.json")))["schro_table_offset_3_8"]
    assert O.quant_offset_3_8() == want


@needs_ref
def test_s16_orc_arithmetic_against_the_compiled_reference_kernels():
    ref = O.reforc()
    qf, qo12 = O.quant_tables()
    qo38 = O.quant_offset_3_8()
    vals = np.arange(-32768, 32768, dtype=np.int32).astype(np.int16)
    for qi in range(61):
        for intra, off in ((1, qo12[qi]), (0, qo38[qi])):
            # in place, any width: orc_dequantise_s16_ip_2d (the general noarith codeblock)
            a = vals.reshape(256, 256).copy()
            ref.orc_dequantise_s16_ip_2d(a.ctypes.data_as(C.c_void_p), a.strides[0], qf[qi], off + 2, 256, 256)
            got = np.zeros((256, 256), np.int16)
            O.dequant_codeblock(got, vals.reshape(256, 256), qi, intra, 1)
            assert np.array_equal(got, a), (qi, intra)
        # the 8- and 4-wide forms read a packed source (schrodecoder.c:3410-3433)
        for n, name in ((8, "orc_dequantise_s16_2d_8xn"), (4, "orc_dequantise_s16_2d_4xn")):
            src = np.ascontiguousarray(vals[1000 * n:1000 * n + 9 * n].reshape(9, n))
            dst = np.zeros((9, 24), np.int16)
            getattr(ref, name)(dst.ctypes.data_as(C.c_void_p), dst.strides[0], src.ctypes.data_as(C.c_void_p),
                               src.strides[0], qf[qi], qo38[qi] + 2, 9)
            got = np.zeros((9, 24), np.int16)
            O.dequant_codeblock(got[:, :n], src, qi, 0, 1)
            assert np.array_equal(got, dst), (qi, name)


@needs_ref
def test_s32_arithmetic_against_the_compiled_reference_kernel():
    ref = O.reforc()
    qf, qo12 = O.quant_tables()
    rng = np.random.default_rng(3)
    small = rng.integers(-70000, 70000, (64, 96)).astype(np.int32)
    small[0, :8] = [0, 1, -1, 2, -2, 32767, -32768, 65536]
    for qi in range(0, 61, 3):
        a = small.copy()
        ref.orc_dequantise_s32_ip_2d(a.ctypes.data_as(C.c_void_p), a.strides[0], qf[qi], qo12[qi] + 2, 96, 64)
        got = np.zeros((64, 96), np.int32)
        O.dequant_codeblock(got, small, qi, 1, 0)
        assert np.array_equal(got, a), qi


def test_c_int_arithmetic_is_schro_dequantise():
    # the arithmetic-coded path: v = (offset + factor * |q| + 2) >> 2 with the sign put back
    # (schrodecoder.c:3072-3079), stored into int16_t
    qf, _ = O.quant_tables()
    qo38 = O.quant_offset_3_8()
    q = np.arange(-3000, 3000, dtype=np.int32).reshape(60, 100)
    for qi in (0, 1, 7, 23, 40, 60):
        mag = (qo38[qi] + qf[qi] * np.abs(q).astype(np.int64) + 2) >> 2
        want = np.where(q == 0, 0, np.sign(q) * mag).astype(np.int64)
        got = np.zeros((60, 100), np.int16)
        O.dequant_codeblock(got, q, qi, 0, 0)
        assert np.array_equal(got, want.astype(np.int16)), qi       # int16_t store truncates


def dequantise_picture(rec, depth, intra):
    """quantised planes + codeblock records -> coefficient planes, the oracle's way."""
    import dirac_stream as D
    out = []
    for comp in range(3):
        q = rec["quant"][comp]
        plane = np.full(q.shape, 0x5a5a, np.int16)
        for (index, x0, y0, x1, y1, zero, qi) in rec["codeblocks"][comp]:
            band = D.subband_view(plane, depth, index)
            qband = D.subband_view(q, depth, index)
            O.dequant_codeblock(band[y0:y1, x0:x1], None if zero else qband[y0:y1, x0:x1], qi, intra, 0)
        if intra:
            ll = D.subband_view(plane, depth, 0)
            ll[...] = O.dc_predict(ll)
        out.append(plane)
    return out


def test_stream_quantised_values_give_the_stream_coefficients():
    # first pictures of the reference's test stream (an intra picture and inter pictures)
    n = 0
    for rec in S.decode_stream(S.load_stream(), S.load_tables(), limit=3, quantised=True):
        if rec["zero_residual"]:
            continue
        got = dequantise_picture(rec, rec["depth"], rec["num_refs"] == 0)
        for comp in range(3):
            assert np.array_equal(got[comp], rec["coeffs"][comp]), (rec["number"], comp)
        # every sample belongs to exactly one codeblock: nothing of the fill survives
        n += 1
    assert n >= 2
