"""CPU: the OBMC oracle (a restatement of schro_motion_render_u8: block scatter, edge /
interior arithmetic, aprons, block-position clamp) against an independent per-pixel
"spec style" model (the shape of schromotionref.c:40-242 and of the GPU kernel): sum of
<= 4 covering blocks, spec weights, per-sample coordinate clamp.  The two must agree
whenever the weighted prediction cannot exceed 255; for gain > 1
the reference itself is inconsistent and only the scatter form is normative."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import synth


def ramp(x, off):
    return (3 if x == 0 else 5) if off == 1 else 1 + (6 * x + off - 1) // (2 * off - 1)


def weights(blen, off):
    w = []
    for i in range(blen):
        if off == 0:
            w.append(8)
        elif i < 2 * off:
            w.append(ramp(i, off))
        elif blen - 1 - i < 2 * off:
            w.append(ramp(blen - 1 - i, off))
        else:
            w.append(8)
    return w


def hp_image(pic):
    up = O.UpComp(pic)
    h, w = pic.shape
    hp = np.zeros((2 * h, 2 * w), np.int64)
    for i in range(4):
        hp[(i >> 1)::2, (i & 1)::2] = up.plane(i)
    return hp


def gather_model(mv, P, k, r1, r2, res, w, h):
    hs, vs = (P["chroma_h_shift"], P["chroma_v_shift"]) if k else (0, 0)
    xbsep, ybsep = P["xbsep_luma"] >> hs, P["ybsep_luma"] >> vs
    xblen, yblen = P["xblen_luma"] >> hs, P["yblen_luma"] >> vs
    xoff, yoff = (xblen - xbsep) // 2, (yblen - ybsep) // 2
    nbx, nby, prec = P["x_num_blocks"], P["y_num_blocks"], P["mv_precision"]
    w1, w2, bits = P["picture_weight_1"], P["picture_weight_2"], P["picture_weight_bits"]
    wx, wy = weights(xblen, xoff), weights(yblen, yoff)
    hp = [hp_image(r1), hp_image(r2)]
    acc = np.zeros((h, w), np.int64)
    cover = np.zeros((h, w), np.int64)
    ys, xs = np.mgrid[0:yblen, 0:xblen]
    for j in range(nby):
        for i in range(nbx):
            m = mv[j * nbx + i]
            mode = int(m["flags"]) & 3
            bx, by = xbsep * i - xoff, ybsep * j - yoff
            px, py = bx + xs, by + ys
            ok = (px >= 0) & (px < w) & (py >= 0) & (py < h)
            if not ok.any():
                continue
            vals = []
            for r in range(2):
                if not mode & (r + 1):
                    vals.append(None)
                    continue
                dx, dy = int(m["v"][r]) >> hs, int(m["v"][2 + r]) >> vs
                x8 = ((px << prec) + dx) << (3 - prec)
                y8 = ((py << prec) + dy) << (3 - prec)
                hx, hy, rx, ry = x8 >> 2, y8 >> 2, x8 & 3, y8 & 3
                cx = lambda v: np.clip(v, 0, 2 * w - 2)
                cy = lambda v: np.clip(v, 0, 2 * h - 2)
                p00, p01 = hp[r][cy(hy), cx(hx)], hp[r][cy(hy), cx(hx + 1)]
                p10, p11 = hp[r][cy(hy + 1), cx(hx)], hp[r][cy(hy + 1), cx(hx + 1)]
                vals.append(((4 - ry) * ((4 - rx) * p00 + rx * p01) + ry * ((4 - rx) * p10 + rx * p11) + 8) >> 4)
            if mode == 0:
                pred = np.full((yblen, xblen), int(m["v"][k]) + 128)
            elif mode == 3:
                pred = (w1 * vals[0] + w2 * vals[1] + ((1 << bits) >> 1)) >> bits
            else:
                pred = ((w1 + w2) * vals[mode - 1] + ((1 << bits) >> 1)) >> bits
            wgt = np.outer(wy, wx)
            np.add.at(acc, (py[ok], px[ok]), (pred * wgt)[ok])
            np.add.at(cover, (py[ok], px[ok]), wgt[ok])
    return acc, cover


@pytest.mark.parametrize("chroma", [(0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("prec", [0, 1, 2, 3])
@pytest.mark.parametrize("blk", [(8, 4), (12, 8), (16, 12), (24, 16)])
def test_scatter_equals_gather_interior(blk, prec, chroma):
    # Away from the picture rim every pixel is covered with total weight 64 and the two
    # formulations must agree exactly; the rim (weight folding) is checked separately below.
    w, h = 96, 64
    for weights3, mv_range in (((1, 1, 1), 80 << prec), ((3, 5, 3), 6 << prec), ((1, 2, 2), 80 << prec)):
        P = synth.motion_params(w, h, blk[0], blk[1], prec, weights3, chroma)
        mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], mv_range, seed=prec + blk[0])
        for k in (0, 1):
            cw = w if k == 0 else -(-w // (1 << chroma[0]))
            ch = h if k == 0 else -(-h // (1 << chroma[1]))
            r1, r2 = synth.picture_u8(ch, cw, seed=5 + k), synth.picture_u8(ch, cw, seed=9 + k)
            res = synth.image_s(ch, cw, np.int16, seed=4)
            got = O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(r1, upsample=prec > 0),
                                  O.UpComp(r2, upsample=prec > 0), res, cw, ch)
            acc, cover = gather_model(mv, P, k, r1, r2, res, cw, ch)
            full = cover == 64
            assert full.mean() > 0.5
            want = np.clip(res.astype(np.int64) + ((acc + 32) >> 6), 0, 255)
            assert np.array_equal(got[full], want[full].astype(np.uint8)), (blk, prec, chroma, weights3, k)


def test_rim_weights_fold_to_64():
    # constant references and zero residual: every output pixel, rim included, must come out as
    # the constant, which is only true if the folded weights sum to 64 everywhere
    w, h = 100, 70
    for blk in ((8, 4), (12, 8), (16, 12), (24, 16)):
        P = synth.motion_params(w, h, blk[0], blk[1], 0, (1, 1, 1), (1, 1))
        mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 8, seed=1, modes=(0, .4, .3, .3))
        for k in (0, 1):
            cw, ch = (w, h) if k == 0 else (w // 2, h // 2)
            r = np.full((ch, cw), 77, np.uint8)
            out = O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(r, upsample=False),
                                  O.UpComp(r, upsample=False), np.zeros((ch, cw), np.int16), cw, ch)
            assert (out == 77).all(), (blk, k)


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built")
def test_block_arithmetic_matches_reference_kernels():
    """orc_combine4_nxm_u8, orc_combine2_nxm_u8, orc_avg2_nxm_u8, orc_rrshift6_add_s16_2d from
    the reference's compiled kernels vs the formulas the oracle and the GPU kernel use."""
    L = O.reforc()
    rng = np.random.default_rng(7)
    n, m = 24, 16
    a, b, c, d = (rng.integers(0, 256, (m, n), dtype=np.uint8) for _ in range(4))
    ai, bi_, ci, di = (x.astype(np.int64) for x in (a, b, c, d))
    vp = C.c_void_p
    ptr = lambda x: x.ctypes.data_as(vp)
    for ry in range(4):
        for rx in range(4):
            w = [(4 - ry) * (4 - rx), (4 - ry) * rx, ry * (4 - rx), ry * rx]
            out = np.zeros((m, n), np.uint8)
            L.orc_combine4_nxm_u8(ptr(out), n, ptr(a), n, ptr(b), n, ptr(c), n, ptr(d), n,
                                  w[0], w[1], w[2], w[3], n, m)
            want = (w[0] * ai + w[1] * bi_ + w[2] * ci + w[3] * di + 8) >> 4
            assert np.array_equal(out, want.astype(np.uint8))
    out = np.zeros((m, n), np.uint8)
    L.orc_avg2_nxm_u8(ptr(out), n, ptr(a), n, ptr(b), n, n, m)
    assert np.array_equal(out, ((ai + bi_ + 1) >> 1).astype(np.uint8))
    for (w1, w2, bits) in ((2, 3, 1), (3, 5, 3), (3, -1, 1), (100, 100, 2)):
        L.orc_combine2_nxm_u8(ptr(out), n, ptr(a), n, ptr(b), n, w1, w2, (1 << bits) >> 1, bits, n, m)
        t = ((a.astype(np.int64) * w1).astype(np.int16).astype(np.int64) + (b.astype(np.int64) * w2).astype(np.int16))
        t = ((t.astype(np.int16).astype(np.int64) + ((1 << bits) >> 1)).astype(np.int16).astype(np.int64)) >> bits
        assert np.array_equal(out, np.clip(t, 0, 255).astype(np.uint8)), (w1, w2, bits)
    res = rng.integers(-32768, 32768, (m, n)).astype(np.int16)
    acc = rng.integers(-32768, 32768, (m, n)).astype(np.int16)
    L.orc_rrshift6_add_s16_2d(ptr(out), n, ptr(res), 2 * n, ptr(acc), 2 * n, n, m)
    t = ((acc.astype(np.int64) + 32).astype(np.int16).astype(np.int64)) >> 6
    t = (res.astype(np.int64) + t).astype(np.int16)
    assert np.array_equal(out, np.clip(t, 0, 255).astype(np.uint8))
    res32 = rng.integers(-2**31, 2**31, (m, n)).astype(np.int32)
    L.orc_rrshift6_add_s32_2d(ptr(out), n, ptr(res32), 4 * n, ptr(acc), 2 * n, n, m)
    t = ((acc.astype(np.int64) + 32).astype(np.int16).astype(np.int64)) >> 6
    t = (res32.astype(np.int16).astype(np.int64) + t).astype(np.int16)      # convlw truncates
    assert np.array_equal(out, np.clip(t, 0, 255).astype(np.uint8))


def test_rrshift6_s16_and_frame_add_against_the_reference_kernels():
    """r06: the two restatements the S16-destination render and schro_hipframe_add are checked with are the reference's
    own compiled kernels where oracle/_ref exists (orc_rrshift6_s16_ip_2d, orc_add_s16_2d, orc_add_s16_u8_2d); the
    numpy forms (what a box without _ref falls back to) give the same bytes, full range included."""
    if not O.ref_available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(5)
    acc = rng.integers(-32768, 32768, (37, 53)).astype(np.int16)
    acc[0, :8] = [-32768, 32767, 8159, 8160, 8161, -24609, -24608, 0]
    assert np.array_equal(O.rrshift6_s16(acc), O.rrshift6_s16_restated(acc))
    # legal sums: (acc + 32) >> 6 - 128
    legal = rng.integers(0, 255 * 64 + 1, (16, 24)).astype(np.int16)
    assert np.array_equal(O.rrshift6_s16(legal), ((legal.astype(np.int32) + 32) >> 6) - 128)
    d = rng.integers(-32768, 32768, (40, 64)).astype(np.int16)
    for src in (rng.integers(-32768, 32768, (33, 70)).astype(np.int16), rng.integers(0, 256, (44, 60)).astype(np.uint8)):
        assert np.array_equal(O.frame_add(d, src), O.frame_add_restated(d, src))
