"""GPU parity: the combine form (r04) -- the OBMC launch writes its PREDICTION, the inverse wavelet's last step adds
it and writes the picture; the residual picture never exists in memory.

  reference:  schro_motion_render (motion, mc_tmp, frame, add = TRUE, output) = orc_rrshift6_add_s16_2d / _s32_2d
              over the inverse transform's output (schromotion8.c:852-876; the stages of the reference's GPU paths:
              x_render_motion -> mc_tmp_frame, x_combine adds, schrodecoder.c:1742-1760, :1908-1921), and
              schro_frame_convert (+ 128) for pictures without references (:1788-1790)
  here:       schro_hip_obmc_batch (prediction_only) -> u8 prediction; schro_hip_iiwt_batch (combine 1 | 2)

Checked against the oracle's two-step result (inverse transform, then motion render with that residual) on every
route: the register kernel's combine epilogue (s16, filters 0-4 and 6, aligned planes), the fallback through a
residual plane in the scratch (s32, the fidelity filter, unaligned / tiny planes), pictures smaller than the
transform (crop), chroma from pair images, and the frame-layer calls.  DC values outside 8 bits are refused loudly."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import schroedinger_amd as sa
import synth
from schroedinger_amd import _lib, frames

pytestmark = pytest.mark.gpu


def up(v, depth):
    return -(-v // (1 << depth)) * (1 << depth)


def run_case(ctx, w, h, depth, filt, dtype=np.int16, chroma=(1, 1), prec=2, blk=(12, 8), seed=1, intra=False, edit_mv=None,
             src_pad=0, weights=(1, 1, 1)):
    hs, vs = chroma
    dims = [(h, w), (-(-h >> vs), -(-w >> hs)), (-(-h >> vs), -(-w >> hs))]
    iw = [(up(ph, depth), up(pw, depth)) for (ph, pw) in dims]
    resid = [(synth.image_s(ih, iwd, dtype, seed=seed + k).astype(np.int64) * 3).astype(dtype) for k, (ih, iwd) in enumerate(iw)]
    coeffs = [O.forward_iwt(r, depth, filt) for r in resid]
    res_want = [O.inverse_iwt(c, depth, filt) for c in coeffs]
    d_co = [ctx.upload(c) for c in coeffs]
    outs = [ctx.plane(ph, pw, np.uint8).fill(0x5e) for (ph, pw) in dims]
    keep = list(d_co) + list(outs)
    if intra:
        ctx.iiwt_batch([(d_co[k], outs[k], None) for k in range(3)], depth, filt)
        for k, (ph, pw) in enumerate(dims):
            assert np.array_equal(outs[k].download(), O.convert_u8(res_want[k], pw, ph)), (k, "intra")
        [p.free() for p in keep]
        return
    P = synth.motion_params(w, h, blk[0], blk[1], prec, weights, chroma)
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 20 << prec, seed + 5)
    if edit_mv is not None:
        edit_mv(mv)
    d_mv = ctx.upload_bytes(mv)
    refs_np = [[synth.picture_u8(ph, pw, seed=seed + 10 * (r + 1) + k) for k, (ph, pw) in enumerate(dims)] for r in range(2)]
    pair = prec > 0 and hs == 1
    hp = []
    for r in range(2):
        if prec == 0:
            planes = [ctx.upload(p) for p in refs_np[r]]
            hp.append(planes)
            keep += planes
            continue
        g0 = ctx.hp_plane(*dims[0])
        ctx.upsample_batch([(ctx.upload(refs_np[r][0]), g0)])
        if pair:
            gp = ctx.hp_plane(*dims[1], pair=True)
            ctx.upsample_batch([((ctx.upload(refs_np[r][1]), ctx.upload(refs_np[r][2])), gp)])
            hp.append([g0, gp, gp])
            keep += [g0, gp]
        else:
            g1, g2 = ctx.hp_plane(*dims[1]), ctx.hp_plane(*dims[2])
            ctx.upsample_batch([(ctx.upload(refs_np[r][1]), g1), (ctx.upload(refs_np[r][2]), g2)])
            hp.append([g0, g1, g2])
            keep += [g0, g1, g2]
    preds = [ctx.plane(ph, pw, np.uint8).fill(0xa1) for (ph, pw) in dims]
    keep += preds + [d_mv]
    ctx.obmc_batch([sa.obmc_plane(d_mv, P, k, hp[0][k], hp[1][k], None, preds[k], prediction_only=True) for k in range(3)])
    ctx.iiwt_batch([(d_co[k], outs[k], preds[k]) for k in range(3)], depth, filt)
    ctx.synchronize()
    for k, (ph, pw) in enumerate(dims):
        want = O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(refs_np[0][k], upsample=prec > 0),
                               O.UpComp(refs_np[1][k], upsample=prec > 0), res_want[k], pw, ph)
        got = outs[k].download()
        if not np.array_equal(got, want):
            bad = np.argwhere(got != want)
            raise AssertionError("component %d: %d mismatches, first at %s" % (k, len(bad), tuple(bad[0])))
    [p.free() for p in keep]


@pytest.mark.parametrize("filt", [0, 1, 2, 3, 4, 6])
def test_register_form_filters(ctx, filt):
    for (w, h, depth) in [(320, 240, 3), (176, 144, 2), (640, 368, 4), (1920, 1080, 3)]:
        run_case(ctx, w, h, depth, filt, seed=filt + 3)


def test_sizes_that_crop_and_ragged_edges(ctx):
    # pictures smaller than the transform (1080 rows in 1088; widths that are not multiples of 8 or 16), tiny planes
    for (w, h) in [(100, 70), (97, 61), (64, 36), (130, 18), (72, 132), (500, 260), (1000, 120), (24, 16)]:
        run_case(ctx, w, h, 2, 0, seed=w)
        run_case(ctx, w, h, 3, 1, chroma=(1, 0), prec=1, seed=h)


def test_fallback_through_a_residual_plane(ctx):
    # what the register kernel's epilogue does not take: s32 coefficients (the Haar one-pass kernel too), the 8-tap
    # fidelity filter, depth 1
    run_case(ctx, 320, 240, 3, 0, dtype=np.int32)
    run_case(ctx, 320, 256, 3, 3, dtype=np.int32)
    run_case(ctx, 320, 240, 2, 5)
    run_case(ctx, 320, 240, 1, 0)
    run_case(ctx, 208, 112, 3, 6, chroma=(0, 0), prec=3, blk=(16, 12))
    run_case(ctx, 208, 112, 3, 2, chroma=(1, 1), prec=0)


@pytest.mark.parametrize("filt,dtype", [(0, np.int16), (1, np.int16), (3, np.int32), (5, np.int16)])
def test_pictures_without_references(ctx, filt, dtype):
    for (w, h, depth) in [(176, 144, 3), (97, 61, 2), (1920, 1080, 3)]:
        run_case(ctx, w, h, depth, filt, dtype=dtype, intra=True)


def test_headline_size(ctx):
    run_case(ctx, 3840, 2160, 3, 0, seed=2)


def test_predictions_outside_8_bits_are_refused(ctx):
    # a DC value outside [-128, 127]: the reference's 16-bit arithmetic wraps; the u8 prediction plane cannot carry
    # it.  Plane layer (the vectors are on the device): the launch raises a flag, the next synchronisation answers
    # SCHRO_HIP_ENEEDS_RESIDUAL and names the prediction_only batch (the residual form stays exact)
    def widen(mv):
        dc = np.flatnonzero((mv["flags"] & 3) == 0)
        mv["v"][dc[::3], :3] = 300
    before = ctx.lib.schro_hip_obmc_prediction_epoch(ctx.h)
    with pytest.raises(sa.SchroHipError, match="does not fit") as ei:
        run_case(ctx, 320, 240, 3, 0, edit_mv=widen)
    assert ei.value.code == _lib.ENEEDS_RESIDUAL
    epoch = ctx.lib.schro_hip_obmc_prediction_epoch(ctx.h)
    assert epoch == before + 1 and ("batch(es) %d " % epoch) in str(ei.value)
    ctx.synchronize()           # (reported once)
    # r06: ... and the batch's number stays fetchable until it has been fetched
    got = (C.c_uint * 4)()
    assert ctx.lib.schro_hip_obmc_overflowed(ctx.h, got, 4) == 1 and got[0] == epoch
    assert ctx.lib.schro_hip_obmc_overflowed(ctx.h, got, 4) == 0
    # ... and weights whose prediction can leave 8 bits are refused at the call
    P = synth.motion_params(96, 64, 12, 8, 2, (2, 3, 1), (1, 1))
    d_mv = ctx.upload_bytes(synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 8, 1))
    g, out = ctx.hp_plane(64, 96), ctx.plane(64, 96, np.uint8)
    with pytest.raises(sa.SchroHipError, match="prediction_only"):
        ctx.obmc_batch([sa.obmc_plane(d_mv, P, 0, g, g, None, out, prediction_only=True)])


@pytest.mark.parametrize("case", ["dc", "gain"])
def test_the_frame_layer_routes_such_pictures_before_it_launches(ctx, case):
    """r05: schro_motion_render_hip (add = FALSE) looks at the host's vectors / weights and answers
    SCHRO_HIP_ENEEDS_RESIDUAL BEFORE launching -- with stage completion OFF and abort-on-error ON (INTEGRATION 2's
    settings: a deferred flag would have surfaced in a later picture's call, or aborted the process) -- and the picture
    decoded in the residual order (inverse transform -> residual frame, motion render with add = TRUE) is the oracle's,
    wrap-around and all (schromotion8.c:542-568, schrodecoder.c:1742-1760)."""
    w, h, hs, vs, prec, filt, depth = 320, 240, 1, 1, 2, 0, 3
    lib = ctx.lib
    pd = [(h, w), (h >> vs, w >> hs), (h >> vs, w >> hs)]
    iw = [(up(ph, depth), up(pw, depth)) for (ph, pw) in pd]
    weights = (1, 1, 1) if case == "dc" else (2, 3, 1)
    P = synth.motion_params(w, h, 12, 8, prec, weights, (hs, vs))
    params = frames.make_params(wavelet_filter_index=filt, transform_depth=depth, iwt_luma_width=iw[0][1], iwt_luma_height=iw[0][0],
                                iwt_chroma_width=iw[1][1], iwt_chroma_height=iw[1][0], num_refs=2,
                                **{k: P[k] for k in ("xblen_luma", "yblen_luma", "xbsep_luma", "ybsep_luma", "mv_precision",
                                                     "picture_weight_bits", "picture_weight_1", "picture_weight_2", "x_num_blocks", "y_num_blocks")})
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 24 << prec, seed=4)
    if case == "dc":
        dc = np.flatnonzero((mv["flags"] & 3) == 0)
        assert len(dc) > 3
        mv["v"][dc[1::2], 0] = 300          # luma DC beyond 8 bits ...
        mv["v"][dc[0::2], 2] = -200         # ... and a V one below
    resid = [synth.image_s(ih, iwd, np.int16, seed=20 + k) for k, (ih, iwd) in enumerate(iw)]
    coeffs = [O.forward_iwt(r, depth, filt) for r in resid]
    res_want = [O.inverse_iwt(c, depth, filt) for c in coeffs]
    fmt16, fmt8 = frames.frame_format(np.int16, hs, vs), frames.frame_format(np.uint8, hs, vs)
    transform_frame = frames.HostFrame(coeffs, hs, vs)
    refs_np = [[synth.picture_u8(ph, pw, seed=40 + 10 * r + k) for k, (ph, pw) in enumerate(pd)] for r in range(2)]
    refs = []
    for r in range(2):
        d = frames.DeviceFrame(ctx, fmt8, w, h).upload(frames.HostFrame(refs_np[r], hs, vs))
        u = frames.DeviceFrame(ctx, fmt8, w, h, upsampled=True)
        sa.check(lib.schro_upsampled_hipframe_upsample(u.ptr(), d.ptr()))
        refs.append(u)
    mc_tmp, out = frames.DeviceFrame(ctx, fmt8, w, h), frames.DeviceFrame(ctx, fmt8, w, h)
    residual = frames.DeviceFrame(ctx, fmt16, iw[0][1], iw[0][0])
    motion = _lib.Motion(refs[0].ptr(), refs[1].ptr(), mv.ctypes.data, C.pointer(params))
    epoch = lib.schro_hip_obmc_prediction_epoch(ctx.h)
    sa.check(lib.schro_hip_context_set_stage_completion(ctx.h, 0))
    lib.schro_hip_set_abort_on_error(1)
    try:
        rc = lib.schro_motion_render_hip(C.byref(motion), mc_tmp.ptr(), None, 0, None)
    finally:
        lib.schro_hip_set_abort_on_error(0)
    assert rc == _lib.ENEEDS_RESIDUAL, rc
    assert b"residual order" in lib.schro_hip_last_error()
    assert lib.schro_hip_obmc_prediction_epoch(ctx.h) == epoch          # nothing was launched
    # the residual order, still without stage completion: the calls only enqueue, one synchronisation at the end
    dev_tf = frames.DeviceFrame(ctx, fmt16, iw[0][1], iw[0][0]).upload(transform_frame)
    sa.check(lib.schro_frame_inverse_iwt_transform_hip(residual.ptr(), dev_tf.ptr(), C.byref(params)))
    sa.check(lib.schro_motion_render_hip(C.byref(motion), None, residual.ptr(), 1, out.ptr()))
    ctx.synchronize()           # (no flag: nothing to report)
    sa.check(lib.schro_hip_context_set_stage_completion(ctx.h, 1))
    got = out.download()
    for k, (ph, pw) in enumerate(pd):
        want = O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(refs_np[0][k]), O.UpComp(refs_np[1][k]), res_want[k], pw, ph)
        assert np.array_equal(got[k], want), k
    for f in (mc_tmp, out, residual, dev_tf) + tuple(refs):
        f.unref()


@pytest.mark.parametrize("hs,vs,prec,filt,depth", [(1, 1, 2, 0, 3), (1, 0, 1, 1, 4), (0, 0, 3, 6, 2)])
def test_frame_layer_calls(ctx, hs, vs, prec, filt, depth):
    """schro_motion_render_hip (motion, mc_tmp, NULL, FALSE, NULL) + schro_frame_inverse_iwt_transform_combine_hip:
    the stage structure of the reference's GPU paths, and the intra form (prediction NULL)."""
    w, h = 320, 240
    lib = ctx.lib
    pd = [(h, w), (-(-h >> vs), -(-w >> hs)), (-(-h >> vs), -(-w >> hs))]
    iw = [(up(ph, depth), up(pw, depth)) for (ph, pw) in pd]
    P = synth.motion_params(w, h, 12, 8, prec, (1, 1, 1), (hs, vs))
    params = frames.make_params(wavelet_filter_index=filt, transform_depth=depth, iwt_luma_width=iw[0][1], iwt_luma_height=iw[0][0],
                                iwt_chroma_width=iw[1][1], iwt_chroma_height=iw[1][0], num_refs=2,
                                **{k: P[k] for k in ("xblen_luma", "yblen_luma", "xbsep_luma", "ybsep_luma", "mv_precision",
                                                     "picture_weight_bits", "picture_weight_1", "picture_weight_2", "x_num_blocks", "y_num_blocks")})
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 24 << prec, seed=4)
    resid = [synth.image_s(ih, iwd, np.int16, seed=20 + k) for k, (ih, iwd) in enumerate(iw)]
    coeffs = [O.forward_iwt(r, depth, filt) for r in resid]
    res_want = [O.inverse_iwt(c, depth, filt) for c in coeffs]
    fmt16, fmt8 = frames.frame_format(np.int16, hs, vs), frames.frame_format(np.uint8, hs, vs)
    transform_frame = frames.HostFrame(coeffs, hs, vs)
    refs_np = [[synth.picture_u8(ph, pw, seed=40 + 10 * r + k) for k, (ph, pw) in enumerate(pd)] for r in range(2)]
    refs = []
    for r in range(2):
        d = frames.DeviceFrame(ctx, fmt8, w, h).upload(frames.HostFrame(refs_np[r], hs, vs))
        if prec > 0:
            u = frames.DeviceFrame(ctx, fmt8, w, h, upsampled=True)
            sa.check(lib.schro_upsampled_hipframe_upsample(u.ptr(), d.ptr()))
            refs.append(u)
        else:
            refs.append(d)
    mc_tmp, out = frames.DeviceFrame(ctx, fmt8, w, h), frames.DeviceFrame(ctx, fmt8, w, h)
    motion = _lib.Motion(refs[0].ptr(), refs[1].ptr(), mv.ctypes.data, C.pointer(params))
    sa.check(lib.schro_motion_render_hip(C.byref(motion), mc_tmp.ptr(), None, 0, None))
    sa.check(lib.schro_frame_inverse_iwt_transform_combine_hip(out.ptr(), transform_frame.ptr(), C.byref(params), mc_tmp.ptr()))
    got = out.download()
    for k, (ph, pw) in enumerate(pd):
        want = O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(refs_np[0][k], upsample=prec > 0),
                               O.UpComp(refs_np[1][k], upsample=prec > 0), res_want[k], pw, ph)
        assert np.array_equal(got[k], want), k
    sa.check(lib.schro_frame_inverse_iwt_transform_combine_hip(out.ptr(), transform_frame.ptr(), C.byref(params), None))
    got = out.download()
    for k, (ph, pw) in enumerate(pd):
        assert np.array_equal(got[k], O.convert_u8(res_want[k], pw, ph)), (k, "intra")
    for f in (mc_tmp, out) + tuple(refs):
        f.unref()


def test_picture_weights(ctx):
    """Weighted prediction (schromotion8.c:542-657, picture_weight_1 / _2 / _bits): weights whose prediction still fits 8 bits take
    the combine form on the item kernel; weights with gain (w1 + w2 > 1 << bits) are refused loudly -- such pictures keep the
    residual form."""
    for weights in ((3, 5, 3), (1, 2, 2), (0, 4, 2)):
        for prec in range(4):
            run_case(ctx, 208, 112, 3, 0, chroma=(1, 1), prec=prec, seed=5 + prec, weights=weights)
    # (r06: a fade on the other block sets -- their weighted row kernels serve the prediction-only launches too)
    for n, blk in enumerate(((8, 4), (16, 12), (24, 16), (16, 8), (24, 12), (32, 16), (20, 12), (28, 16))):
        for prec in range(4):
            run_case(ctx, 208, 112, 3, 0, chroma=[(1, 1), (0, 0), (1, 0)][n % 3], prec=prec, blk=blk, seed=9 + prec, weights=(3, 5, 3))
    with pytest.raises(sa.SchroHipError, match="prediction_only"):
        run_case(ctx, 208, 112, 3, 0, chroma=(1, 1), prec=2, seed=5, weights=(2, 3, 1))


def test_depth_one_call_with_an_unaligned_ll_plane_takes_the_scratch_route(ctx):
    """r05: the second call of a transform in two calls (depth 1, SchroHipIwtPlane.ll) in the combine form,
    with an LL plane whose rows are NOT 8-byte aligned: the register kernel cannot read it, and the call must fall back to
    the residual plane in the scratch + convert route (it returned EINVAL, "promised a register tile") -- same picture."""
    from schroedinger_amd import DevicePlane
    h, w, depth, filt = 544, 960, 3, 0
    img = (synth.image_s(h, w, np.int16, seed=31).astype(np.int64) * 3).astype(np.int16)
    co = O.forward_iwt(img, depth, filt)
    res_want = O.inverse_iwt(co, depth, filt)
    d_co = ctx.upload(co)
    ll_al = ctx.plane(h // 2, w // 2, np.int16)
    ctx.iiwt_batch([(d_co.level_view(1), ll_al)], depth - 1, filt)
    # the same LL band two bytes into a wider plane: pointer % 8 == 2
    wide = ctx.plane(h // 2, w // 2 + 8, np.int16)
    host = np.zeros((h // 2, w // 2 + 8), np.int16)
    host[:, 1:1 + w // 2] = ll_al.download()
    wide.upload(host)
    ll_un = DevicePlane.__new__(DevicePlane)
    ll_un.ctx, ll_un.dtype = ctx, np.dtype(np.int16)
    ll_un.height, ll_un.width, ll_un.stride = h // 2, w // 2, wide.stride
    ll_un.nbytes, ll_un.ptr = 0, wide.ptr + 2
    for ll in (ll_al, ll_un):
        out = ctx.plane(h, w, np.uint8).fill(0x33)
        ctx.iiwt_batch([(d_co, out, None)], 1, filt, ll=[ll])
        assert np.array_equal(out.download(), O.convert_u8(res_want, w, h)), "aligned" if ll is ll_al else "unaligned"
        out.free()
    for p in (d_co, ll_al, wide):
        p.free()


def test_predictions_alone_over_random_geometries(ctx):
    """r05: the prediction-only row kernels store (sum + 32) >> 6 without a clamp where no DC value is wide -- the weights
    over a pixel add up to 64 wherever it lies, folded rims included.  Random sizes, block sets, precisions and chroma
    formats; the u8 prediction itself against the oracle's render of a zero residual."""
    rng = np.random.default_rng(77)
    blocks = [(8, 4), (12, 8), (16, 8), (16, 12), (8, 8), (4, 4)]
    for n in range(24):
        w, h = int(rng.integers(3, 40)) * 8 + int(rng.integers(0, 8)), int(rng.integers(3, 30)) * 4 + int(rng.integers(0, 4))
        xblen, xbsep = blocks[int(rng.integers(0, len(blocks)))]
        prec = int(rng.integers(1, 3))
        chroma = [(1, 1), (1, 0), (0, 0)][int(rng.integers(0, 3))]
        P = synth.motion_params(w, h, xblen, xbsep, prec, (1, 1, 1), chroma)
        mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 24 << prec, 300 + n)
        d_mv = ctx.upload_bytes(mv)
        dims = [(h, w)] + [(-(-h // (1 << chroma[1])), -(-w // (1 << chroma[0])))] * 2
        refs_np = [[synth.picture_u8(ph, pw, seed=400 + 10 * r + k + n) for k, (ph, pw) in enumerate(dims)] for r in range(2)]
        pair = chroma[0] == 1
        hp, keep = [], [d_mv]
        for r in range(2):
            g0 = ctx.hp_plane(*dims[0])
            ctx.upsample_batch([(ctx.upload(refs_np[r][0]), g0)])
            if pair:
                gp = ctx.hp_plane(*dims[1], pair=True)
                ctx.upsample_batch([((ctx.upload(refs_np[r][1]), ctx.upload(refs_np[r][2])), gp)])
                hp.append([g0, gp, gp])
                keep += [g0, gp]
            else:
                g1, g2 = ctx.hp_plane(*dims[1]), ctx.hp_plane(*dims[2])
                ctx.upsample_batch([(ctx.upload(refs_np[r][1]), g1), (ctx.upload(refs_np[r][2]), g2)])
                hp.append([g0, g1, g2])
                keep += [g0, g1, g2]
        preds = [ctx.plane(ph, pw, np.uint8).fill(0xa1) for (ph, pw) in dims]
        ctx.obmc_batch([sa.obmc_plane(d_mv, P, k, hp[0][k], hp[1][k], None, preds[k], prediction_only=True) for k in range(3)])
        ctx.synchronize()
        for k, (ph, pw) in enumerate(dims):
            want = O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(refs_np[0][k], upsample=True),
                                   O.UpComp(refs_np[1][k], upsample=True), np.zeros((ph, pw), np.int16), pw, ph)
            got = preds[k].download()
            assert np.array_equal(got, want), (n, w, h, xblen, xbsep, prec, chroma, k, int((got != want).sum()))
        for p in keep + preds:
            p.free()


def test_a_pipelining_host_learns_every_batch_that_overflowed(ctx):
    """r06: twenty prediction_only batches enqueued without a synchronisation in between -- more than the
    ring of flag words holds --, every third one with a DC value outside 8 bits; calls that only enqueue (more OBMC
    batches, a DC prediction, a transform) are never refused for another picture's routing answer; the synchronisation
    at the end names the batches once and schro_hip_obmc_overflowed returns exactly the overflowed ones."""
    w, h = 96, 64
    P = synth.motion_params(w, h, 12, 8, 2, (1, 1, 1), (1, 1))
    g, out = ctx.hp_plane(h, w), ctx.plane(h, w, np.uint8)
    ctx.upsample_batch([(ctx.upload(synth.picture_u8(h, w, seed=3)), g)])
    ctx.synchronize()
    ctx.lib.schro_hip_context_set_stage_completion(ctx.h, 0)
    try:
        base = ctx.lib.schro_hip_obmc_prediction_epoch(ctx.h)
        want, keep = [], []
        band = ctx.upload(synth.image_s(16, 24, np.int16, seed=1))
        for n in range(20):
            mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 8, seed=40 + n, modes=(0.3, 0.3, 0.1, 0.3))
            if n % 3 == 1:
                dc = np.flatnonzero((mv["flags"] & 3) == 0)
                mv["v"][dc[::2], :3] = -700            # (blocks on the rim store their DC as a uint8_t: take many)
                want.append(base + n + 1)
            d_mv = ctx.upload_bytes(mv)
            keep.append(d_mv)
            ctx.obmc_batch([sa.obmc_plane(d_mv, P, 0, g, g, None, out, prediction_only=True)])
            ctx.dc_predict_batch([band])                # an unrelated enqueue: not refused
        with pytest.raises(sa.SchroHipError, match="residual order") as ei:
            ctx.synchronize()
        assert ei.value.code == _lib.ENEEDS_RESIDUAL
        for e in want[:8]:
            assert (" %d" % e) in str(ei.value) or ("%d," % e) in str(ei.value), (e, str(ei.value))
        ctx.synchronize()                               # named once
        got = (C.c_uint * 32)()
        n = ctx.lib.schro_hip_obmc_overflowed(ctx.h, got, 3)
        assert n == 3
        rest = (C.c_uint * 32)()
        m = ctx.lib.schro_hip_obmc_overflowed(ctx.h, rest, 32)
        assert sorted(list(got[:n]) + list(rest[:m])) == want
        assert ctx.lib.schro_hip_obmc_overflowed(ctx.h, rest, 32) == 0
        for p in keep:
            p.free()
    finally:
        ctx.lib.schro_hip_context_set_stage_completion(ctx.h, 1)
