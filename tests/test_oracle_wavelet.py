"""CPU: the wavelet oracle against (a) the reference's own compiled kernels driven in
the reference's row schedule (oracle/_ref), (b) the reference's test design
(testsuite/wavelet_2d.c: perfect reconstruction + a scalar column-then-row model built
on testsuite/common.c synth()), (c) the committed golden vectors."""
import os

import numpy as np
import pytest

import oracle_lib as O
import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SHIFT = [1, 1, 1, 0, 1, 0, 1]          # filtershift[], testsuite/wavelet_2d.c


def ext(a, n):
    """extend() of testsuite/common.c:731-749: same-parity replicate, 8 each side."""
    out = np.empty(n + 16, np.int64)
    out[8:8 + n] = a
    for k in range(1, 9):
        out[8 - k] = a[1] if k % 2 else a[0]
        out[8 + n - 1 + k] = a[n - 2] if k % 2 else a[n - 1]
    return out


def synth_1d(a, filt):
    """synth() of testsuite/common.c:751-827, plain (non-wrapping) integers."""
    a = np.array(a, np.int64)
    n = len(a)

    def step(update):
        e = ext(a, n)
        update(e)
        a[:] = e[8:8 + n]

    ev = np.arange(0, n, 2) + 8

    def sr(x, s):
        return x >> s

    if filt in (0, 1):
        step(lambda e: e.__setitem__(ev, e[ev] - sr(e[ev - 1] + e[ev + 1] + 2, 2)))
        if filt == 0:
            step(lambda e: e.__setitem__(ev + 1, e[ev + 1] + sr(-e[ev - 2] + 9 * e[ev] + 9 * e[ev + 2] - e[ev + 4] + 8, 4)))
        else:
            step(lambda e: e.__setitem__(ev + 1, e[ev + 1] + sr(e[ev] + e[ev + 2] + 1, 1)))
    elif filt == 2:
        step(lambda e: e.__setitem__(ev, e[ev] - sr(-e[ev - 3] + 9 * e[ev - 1] + 9 * e[ev + 1] - e[ev + 3] + 16, 5)))
        step(lambda e: e.__setitem__(ev + 1, e[ev + 1] + sr(-e[ev - 2] + 9 * e[ev] + 9 * e[ev + 2] - e[ev + 4] + 8, 4)))
    elif filt in (3, 4):
        step(lambda e: e.__setitem__(ev, e[ev] - sr(e[ev + 1] + 1, 1)))
        step(lambda e: e.__setitem__(ev + 1, e[ev + 1] + e[ev]))
    elif filt == 5:
        step(lambda e: e.__setitem__(ev + 1, e[ev + 1] + sr(-2 * e[ev - 6] + 10 * e[ev - 4] - 25 * e[ev - 2] + 81 * e[ev]
                                                            + 81 * e[ev + 2] - 25 * e[ev + 4] + 10 * e[ev + 6] - 2 * e[ev + 8] + 128, 8)))
        step(lambda e: e.__setitem__(ev, e[ev] - sr(-8 * e[ev - 7] + 21 * e[ev - 5] - 46 * e[ev - 3] + 161 * e[ev - 1]
                                                    + 161 * e[ev + 1] - 46 * e[ev + 3] + 21 * e[ev + 5] - 8 * e[ev + 7] + 128, 8)))
    else:
        step(lambda e: e.__setitem__(ev, e[ev] - sr(1817 * e[ev - 1] + 1817 * e[ev + 1] + 2048, 12)))
        step(lambda e: e.__setitem__(ev + 1, e[ev + 1] - sr(3616 * e[ev] + 3616 * e[ev + 2] + 2048, 12)))
        step(lambda e: e.__setitem__(ev, e[ev] + sr(217 * e[ev - 1] + 217 * e[ev + 1] + 2048, 12)))
        step(lambda e: e.__setitem__(ev + 1, e[ev + 1] + sr(6497 * e[ev] + 6497 * e[ev + 2] + 2048, 12)))
    return a


def iiwt_ref(p, filt):
    """iiwt_ref() of testsuite/wavelet_2d.c:380-409: columns, then rows, then rshift."""
    p = p.astype(np.int64)
    h, w = p.shape
    for x in range(w):
        p[:, x] = synth_1d(p[:, x], filt)
    for y in range(h):
        row = np.empty(w, np.int64)
        row[0::2], row[1::2] = p[y, : w // 2], p[y, w // 2:]
        p[y] = synth_1d(row, filt)
    if SHIFT[filt]:
        p = (p + 1) >> 1
    return p


@pytest.mark.parametrize("dtype", [np.int16, np.int32])
@pytest.mark.parametrize("filt", range(7))
def test_perfect_reconstruction_size_sweep(filt, dtype):
    # every even size 2..40 x 2..40 like wavelet_2d.c:288-299 (random pattern)
    for w in range(2, 41, 2):
        for h in range(2, 41, 6):
            img = synth.image_s(h, w, dtype, seed=w * 41 + h)
            assert np.array_equal(O.iiwt_2d(O.iwt_2d(img, filt), filt), img), (filt, h, w)


@pytest.mark.parametrize("filt", range(7))
def test_matches_reference_test_model(filt):
    # the scalar model only defines the non-wrapping domain, as in the reference's test
    for (h, w) in [(20, 20), (2, 2), (6, 4), (18, 34), (34, 18)]:
        img = synth.image_s(h, w, np.int16, seed=h + w)
        co = O.iwt_2d(img, filt)
        assert np.array_equal(O.iiwt_2d(co, filt), iiwt_ref(co, filt).astype(np.int16)), (filt, h, w)


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("dtype", [np.int16, np.int32])
@pytest.mark.parametrize("filt", [0, 1, 2, 3, 4, 6])
def test_matches_reference_kernels_in_reference_schedule(filt, dtype):
    for (h, w) in [(2, 2), (4, 6), (6, 4), (16, 16), (18, 34), (48, 64), (2, 40), (40, 2), (256, 256)]:
        co = O.iwt_2d(synth.image_s(h, w, dtype, seed=3), filt)
        assert np.array_equal(O.iiwt_2d(co, filt), O.refdrv_iiwt_2d(co, filt)), (filt, h, w, "legal")
        fr = synth.full_range(h, w, dtype, seed=5)      # pins every wrap point
        assert np.array_equal(O.iiwt_2d(fr, filt), O.refdrv_iiwt_2d(fr, filt)), (filt, h, w, "full")


def test_rounding_and_wrap_pins():
    # (x+1)>>1 wraps at 16 bits for filters 0,1,2,6 (orc_interleave2_rrshift1_s16) but not for
    # Haar1 (avgsw): 32767 -> -16384 vs 16384
    z = np.zeros((2, 2), np.int16)
    z[0, 0] = 32767
    assert O.iiwt_2d(z, 3)[0, 0] == 32767
    assert O.iiwt_2d(z, 4)[0, 0] == 16384
    one = np.zeros((2, 2), np.int16)
    one[0, 0] = 1
    assert O.iiwt_2d(one, 1)[0, 0] == 1         # (1+1)>>1
    assert O.iiwt_2d(-one, 1)[0, 0] == 0        # (-1+1)>>1


def test_level_loop_layout():
    # level view {w>>l, h>>l, stride<<l}: the LL band of level l is the level l+1 view
    img = synth.image_s(32, 48, np.int16, seed=9)
    co = O.forward_iwt(img, 2, 1)
    step = co.copy()
    lvl1 = O.iiwt_2d(np.ascontiguousarray(step[0::2, :24]), 1)
    step[0::2, :24] = lvl1
    assert np.array_equal(O.iiwt_2d(step, 1), O.inverse_iwt(co, 2, 1))


def test_golden_vectors():
    g = np.load(os.path.join(GOLD, "iiwt_oracle.npz"))
    keys = [k[:-3] for k in g.files if k.endswith("_in")]
    assert len(keys) > 100
    for k in keys:
        parts = k.split("_")
        filt = int(parts[1][1:])
        x = g[k + "_in"]
        want = g[k + "_out"]
        got = O.inverse_iwt(x, 3, filt) if "_d3" in k else O.iiwt_2d(x, filt)
        assert np.array_equal(got, want), k
    r = np.load(os.path.join(GOLD, "iiwt_ref_kernels.npz"))
    for k in [k[:-3] for k in r.files if k.endswith("_in")]:
        filt = int(k.split("_")[1][1:])
        assert np.array_equal(O.iiwt_2d(r[k + "_in"], filt), r[k + "_out"]), k
