"""GPU: the SchroFrame-shaped stage boundary (frame layer of include/schro_hip.h) driven the
way the reference's stage bodies drive their GPU back end (schrodecoder.c:1697-2141):
x_wavelet_transform -> x_upsample (of the references) -> x_render_motion -> x_combine,
for an inter picture and an intra picture, 4:2:0 and 4:2:2, checked against the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import schroedinger_amd as sa
import synth
from schroedinger_amd import _lib, frames

pytestmark = pytest.mark.gpu


def dims(w, h, hs, vs):
    cw, ch = -(-w // (1 << hs)), -(-h // (1 << vs))
    return [(h, w), (ch, cw), (ch, cw)]


def round_up(x, n):
    return (x + (1 << n) - 1) >> n << n


@pytest.mark.parametrize("hs,vs,prec,filt,depth", [(1, 1, 2, 0, 3), (1, 0, 0, 1, 4), (0, 0, 1, 6, 2), (1, 1, 3, 3, 3)])
def test_inter_picture_through_stage_calls(ctx, hs, vs, prec, filt, depth):
    w, h = 320, 240
    lib = ctx.lib
    pd = dims(w, h, hs, vs)
    iw = [(round_up(ph, depth), round_up(pw, depth)) for ph, pw in pd]
    P = synth.motion_params(w, h, 12, 8, prec, (1, 1, 1), (hs, vs))
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 48 << prec, seed=4)
    params = frames.make_params(
        wavelet_filter_index=filt, transform_depth=depth, iwt_luma_width=iw[0][1], iwt_luma_height=iw[0][0],
        iwt_chroma_width=iw[1][1], iwt_chroma_height=iw[1][0], num_refs=2, xblen_luma=12, yblen_luma=12,
        xbsep_luma=8, ybsep_luma=8, mv_precision=prec, picture_weight_bits=1, picture_weight_1=1,
        picture_weight_2=1, x_num_blocks=P["x_num_blocks"], y_num_blocks=P["y_num_blocks"])

    # picture->transform_frame on the host (iwt-padded), coefficients from a forward transform
    resid = [synth.image_s(ih, iwd, np.int16, seed=20 + k) for k, (ih, iwd) in enumerate(iw)]
    coeffs = [O.forward_iwt(r, depth, filt) for r in resid]
    transform_frame = frames.HostFrame(coeffs, hs, vs)
    fmt16, fmt8 = frames.frame_format(np.int16, hs, vs), frames.frame_format(np.uint8, hs, vs)

    # x_wavelet_transform: picture->frame = clone(device, transform_frame); inverse transform
    frame = frames.DeviceFrame(ctx, fmt16, iw[0][1], iw[0][0])
    sa.check(lib.schro_frame_inverse_iwt_transform_hip(frame.ptr(), transform_frame.ptr(), C.byref(params)))
    got_res = frame.download()
    for k in range(3):
        assert np.array_equal(got_res[k], O.inverse_iwt(coeffs[k], depth, filt)), k

    # two reference pictures already on the device; x_upsample when mv_precision > 0
    refs_np = [[synth.picture_u8(ph, pw, seed=40 + 10 * r + k) for k, (ph, pw) in enumerate(pd)] for r in range(2)]
    refs = []
    for r in range(2):
        d = frames.DeviceFrame(ctx, fmt8, w, h).upload(frames.HostFrame(refs_np[r], hs, vs))
        if prec > 0:
            u = frames.DeviceFrame(ctx, fmt8, w, h, upsampled=True)
            sa.check(lib.schro_upsampled_hipframe_upsample(u.ptr(), d.ptr()))
            sa.check(lib.schro_upsampled_hipframe_upsample(u.ptr(), d.ptr()))   # second call is a no-op
            hp = u.download()
            # (chroma of 4:2:0 / 4:2:2 frames: one (U, V) pair image behind components[1], r04)
            assert u.c.is_upsampled == (2 if hs else 1)
            for k in range(3):
                upk = O.UpComp(refs_np[r][k])
                for pl in range(4):
                    assert np.array_equal(hp[k][pl >> 1::2, pl & 1::2], upk.plane(pl)), (k, pl)
            refs.append(u)
        else:
            refs.append(d)

    # x_render_motion: schro_motion_render (motion, mc_tmp, frame, add=TRUE, ref_output_frame)
    out = frames.DeviceFrame(ctx, fmt8, w, h)
    motion = _lib.Motion(refs[0].ptr(), refs[1].ptr(), mv.ctypes.data, C.pointer(params))
    sa.check(lib.schro_motion_render_hip(C.byref(motion), None, frame.ptr(), 1, out.ptr()))
    got = out.download()
    for k, (ph, pw) in enumerate(pd):
        want = O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(refs_np[0][k], upsample=prec > 0),
                               O.UpComp(refs_np[1][k], upsample=prec > 0), got_res[k], pw, ph)
        assert np.array_equal(got[k], want), k

    # x_combine: u8 -> u8 copy into the output picture
    outpic = frames.DeviceFrame(ctx, fmt8, w, h)
    sa.check(lib.schro_hipframe_convert(outpic.ptr(), out.ptr()))
    for a, b in zip(outpic.download(), got):
        assert np.array_equal(a, b)
    for f in (frame, out, outpic) + tuple(refs):
        f.unref()


def test_intra_picture_convert(ctx):
    w, h, depth, filt = 176, 144, 3, 0
    lib = ctx.lib
    iw = [(round_up(144, 3), round_up(176, 3)), (round_up(72, 3), round_up(88, 3))]
    iw = [iw[0], iw[1], iw[1]]
    resid = [(synth.image_s(ih, iwd, np.int16, seed=7 + k).astype(np.int32) * 3).astype(np.int16)
             for k, (ih, iwd) in enumerate(iw)]
    coeffs = [O.forward_iwt(r, depth, filt) for r in resid]
    params = frames.make_params(wavelet_filter_index=filt, transform_depth=depth, iwt_luma_width=iw[0][1],
                                iwt_luma_height=iw[0][0], iwt_chroma_width=iw[1][1], iwt_chroma_height=iw[1][0])
    frame = frames.DeviceFrame(ctx, sa.FORMAT_S16_420, iw[0][1], iw[0][0])
    sa.check(lib.schro_frame_inverse_iwt_transform_hip(frame.ptr(), frames.HostFrame(coeffs, 1, 1).ptr(),
                                                       C.byref(params)))
    out = frames.DeviceFrame(ctx, sa.FORMAT_U8_420, w, h)
    sa.check(lib.schro_hipframe_convert(out.ptr(), frame.ptr()))       # schrodecoder.c:1788-1790
    got = out.download()
    for k, (ph, pw) in enumerate([(144, 176), (72, 88), (72, 88)]):
        assert np.array_equal(got[k], O.convert_u8(O.inverse_iwt(coeffs[k], depth, filt), pw, ph))
    frame.unref()
    out.unref()


def test_error_behaviour(ctx):
    lib = ctx.lib
    P = synth.motion_params(64, 48, 12, 8, 2, (1, 1, 1), (1, 1))
    params = frames.make_params(num_refs=1, xblen_luma=12, yblen_luma=12, xbsep_luma=8, ybsep_luma=8,
                                mv_precision=2, picture_weight_bits=1, picture_weight_1=1, picture_weight_2=1,
                                x_num_blocks=P["x_num_blocks"], y_num_blocks=P["y_num_blocks"],
                                have_global_motion=1)
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 8, seed=1)
    ref = frames.DeviceFrame(ctx, sa.FORMAT_U8_420, 64, 48)             # NOT upsampled
    res = frames.DeviceFrame(ctx, sa.FORMAT_S16_420, 64, 48)
    out = frames.DeviceFrame(ctx, sa.FORMAT_U8_420, 64, 48)
    motion = _lib.Motion(ref.ptr(), None, mv.ctypes.data, C.pointer(params))
    assert lib.schro_motion_render_hip(C.byref(motion), None, res.ptr(), 1, out.ptr()) == -4   # global motion
    params.have_global_motion = 0
    assert lib.schro_motion_render_hip(C.byref(motion), None, res.ptr(), 1, out.ptr()) == -1   # plain ref at qpel
    assert b"upsampled" in lib.schro_hip_last_error()
    assert lib.schro_hipframe_convert(res.ptr(), out.ptr()) == -4                       # u8 -> s16 not on path
    frames.DeviceFrame(ctx, sa.FORMAT_U8_420, 64, 48).unref()
    before = ctx.domain_bytes()
    for _ in range(3):      # the domain recycles same-size blocks (schrodomain.c:58-103)
        f = frames.DeviceFrame(ctx, sa.FORMAT_U8_420, 64, 48)
        f.unref()
    assert ctx.domain_bytes() == before
    for f in (ref, res, out):
        f.unref()


def test_copy_out_to_packed_frames(ctx):
    # x_combine's schro_frame_convert (&output_picture, ref_output_frame) when the application
    # supplied a packed frame (schrodecoder.c:2011, 2052): device u8 picture -> device packed
    # frame -> host, half the bytes of the planar download for YUYV / UYVY
    lib = ctx.lib
    for (w, h, fmt8, hs, vs) in [(176, 144, sa.FORMAT_U8_420, 1, 1), (97, 35, sa.FORMAT_U8_422, 1, 0),
                                 (64, 48, sa.FORMAT_U8_444, 0, 0)]:
        cw, ch = -(-w // (1 << hs)), -(-h // (1 << vs))
        pl = [synth.picture_u8(h, w, seed=2), synth.picture_u8(ch, cw, seed=3), synth.picture_u8(ch, cw, seed=4)]
        dev = frames.DeviceFrame(ctx, fmt8, w, h).upload(frames.HostFrame(pl, hs, vs))
        for fmt in (sa.FORMAT_YUYV, sa.FORMAT_UYVY, sa.FORMAT_AYUV):
            for (W, H) in [(w, h), (w + 6, h + 2), (w - 8, h - 4)]:
                out = frames.DeviceFrame(ctx, fmt, W, H)
                sa.check(lib.schro_hipframe_convert(out.ptr(), dev.ptr()))
                assert np.array_equal(out.download(), O.pack_u8(pl, hs, vs, fmt, W, H)), (fmt8, fmt, W, H)
                out.unref()
        dev.unref()


def test_copy_out_of_deep_frames(ctx):
    # the > 8-bit end of x_combine (schrodecoder.c:2005-2021): an intra picture's s16 / s32 frame,
    # shifted down when the stream is deeper than the output picture, then schro_frame_convert
    # into the packed frame the application supplied (v210, v216, ARGB, AY64)
    lib = ctx.lib
    for dtype, fmts in ((np.int16, [(sa.FORMAT_V216, 1), (0x106, 1), (sa.FORMAT_ARGB, 0)]),
                        (np.int32, [(sa.FORMAT_AY64, 0), (sa.FORMAT_V216, 1)])):
        for fmt, hs in fmts:
            w, h = 98, 20
            cw = -(-w >> hs)
            pl = [synth.image_s(h, w, dtype, seed=5) * 3, synth.image_s(h, cw, dtype, seed=6) * 3,
                  synth.image_s(h, cw, dtype, seed=7) * 3]
            dev = frames.DeviceFrame(ctx, frames.frame_format(dtype, hs, 0), w, h).upload(frames.HostFrame(pl, hs, 0))
            sa.check(lib.schro_hipframe_shift_right(dev.ptr(), 2))
            shifted = [O.shift_right(p, 2) for p in pl]
            for a, b in zip(dev.download(), shifted):
                assert np.array_equal(a, b)
            for (W, H) in [(w, h), (w + 6, h + 2), (w - 8, h - 4)]:
                out = frames.DeviceFrame(ctx, fmt, W, H)
                sa.check(lib.schro_hipframe_convert(out.ptr(), dev.ptr()))
                want = O.pack_v210(shifted, hs, 0, W, H) if fmt == 0x106 else O.pack_wide(shifted, hs, 0, W, H, fmt)
                assert np.array_equal(out.download(), want), (dtype, hex(fmt), W, H)
                out.unref()
            dev.unref()


@pytest.mark.parametrize("w,h,dtype,filt,depth", [(96, 48, np.int32, 3, 3), (1008, 40, np.int32, 3, 3), (96, 48, np.int16, 0, 2),
                                                  (100, 36, np.int32, 4, 3)])
def test_transform_and_copy_out_of_an_intra_picture_in_one_stage_call(ctx, w, h, dtype, filt, depth):
    """r05: schro_frame_inverse_iwt_transform_convert_hip (packed, transform_frame, params) -- x_wavelet_transform and the
    schro_frame_convert of x_combine for a picture without references and a v210 output picture (schrodecoder.c:1855-1886,
    :2011-2052) -- equals the two stage calls and the oracle's chain; on the fused kernel (s32 Haar, depth 3, multiples of
    48 x 8) and on the two-pass route (everything else, incl. a picture smaller than the padded transform)."""
    lib = ctx.lib
    iw, ih = -(-w // (1 << depth)) * (1 << depth), -(-h // (1 << depth)) * (1 << depth)
    icw, ich = -(-(-(-w // 2)) // (1 << depth)) * (1 << depth), ih
    if (icw << 1) != iw:        # (keep the chroma transform size = the luma size shifted: what the entry asks for)
        iw = icw << 1
    dims_t = [(ih, iw), (ich, icw), (ich, icw)]
    co = [O.forward_iwt((synth.image_s(a, b, dtype, seed=11 + k).astype(np.int64) * 5).astype(dtype), depth, filt) for k, (a, b) in enumerate(dims_t)]
    params = frames.make_params(wavelet_filter_index=filt, transform_depth=depth, iwt_luma_width=iw, iwt_luma_height=ih,
                                iwt_chroma_width=icw, iwt_chroma_height=ich, num_refs=0)
    fmt = frames.frame_format(dtype, 1, 0)
    tf = frames.DeviceFrame(ctx, fmt, iw, ih).upload(frames.HostFrame(co, 1, 0))
    px = [O.inverse_iwt(c, depth, filt) for c in co]
    want = O.pack_v210([px[0][:h, :w], px[1][:h, :-(-w // 2)], px[2][:h, :-(-w // 2)]], 1, 0, w, h)
    out = frames.DeviceFrame(ctx, 0x106, w, h)
    sa.check(lib.schro_frame_inverse_iwt_transform_convert_hip(out.ptr(), tf.ptr(), C.byref(params)))
    assert np.array_equal(out.download(), want)
    # the two stage calls it replaces
    frame = frames.DeviceFrame(ctx, fmt, w, h) if (w, h) == (iw, ih) else None
    if frame is not None:
        out2 = frames.DeviceFrame(ctx, 0x106, w, h)
        sa.check(lib.schro_frame_inverse_iwt_transform_hip(frame.ptr(), tf.ptr(), C.byref(params)))
        sa.check(lib.schro_hipframe_convert(out2.ptr(), frame.ptr()))
        assert np.array_equal(out2.download(), want)
        frame.unref()
        out2.unref()
    out.unref()
    tf.unref()


@pytest.mark.parametrize("hs,vs,prec", [(1, 1, 2), (1, 0, 1), (0, 0, 0), (1, 1, 3)])
def test_zero_residual_picture_has_no_frame_to_add(ctx, hs, vs, prec):
    """schrodecoder.c:1800, :1861, :1904-1906: a zero_residual picture runs no wavelet stage and its GPU paths
    take the prediction as the combined frame.  schro_motion_render_hip (motion, NULL, NULL, TRUE, out) and a
    plane-layer job without a residual: the prediction alone, clamped = the oracle's with a residual of zeros.
    (r03 required a frame of zeros: 2 bytes per pixel uploaded and read for nothing.)"""
    w, h = 208, 112
    lib = ctx.lib
    pd = dims(w, h, hs, vs)
    P = synth.motion_params(w, h, 12, 8, prec, (1, 1, 1), (hs, vs))
    params = frames.make_params(num_refs=2, **{k: P[k] for k in (
        "xblen_luma", "yblen_luma", "xbsep_luma", "ybsep_luma", "mv_precision", "picture_weight_bits",
        "picture_weight_1", "picture_weight_2", "x_num_blocks", "y_num_blocks")})
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 24 << prec, seed=31)
    fmt8 = frames.frame_format(np.uint8, hs, vs)
    refs_np = [[synth.picture_u8(ph, pw, seed=70 + 10 * r + k) for k, (ph, pw) in enumerate(pd)] for r in range(2)]
    refs = []
    for r in range(2):
        d = frames.DeviceFrame(ctx, fmt8, w, h).upload(frames.HostFrame(refs_np[r], hs, vs))
        if prec > 0:
            u = frames.DeviceFrame(ctx, fmt8, w, h, upsampled=True)
            sa.check(lib.schro_upsampled_hipframe_upsample(u.ptr(), d.ptr()))
            refs.append(u)
        else:
            refs.append(d)
    out = frames.DeviceFrame(ctx, fmt8, w, h)
    motion = _lib.Motion(refs[0].ptr(), refs[1].ptr(), mv.ctypes.data, C.pointer(params))
    sa.check(lib.schro_motion_render_hip(C.byref(motion), None, None, 1, out.ptr()))
    got = out.download()
    wants = []
    for k, (ph, pw) in enumerate(pd):
        wants.append(O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(refs_np[0][k], upsample=prec > 0),
                                     O.UpComp(refs_np[1][k], upsample=prec > 0), np.zeros((ph, pw), np.int16), pw, ph))
        assert np.array_equal(got[k], wants[k]), k
    # the plane layer: residual NULL, with each of the kernels the precision / geometry selects
    d_mv = ctx.upload_bytes(mv)
    c = refs[0].c.components
    jobs, outs = [], []
    for k, (ph, pw) in enumerate(pd):
        class V:
            pass
        views = []
        for fr in refs:
            v = V()
            v.ptr, v.stride = fr.c.components[k].data, fr.c.components[k].stride
            v.pair = k > 0 and fr.c.is_upsampled == 2
            views.append(v)
        o = ctx.plane(ph, pw, np.uint8).fill(9)
        jobs.append(sa.obmc_plane(d_mv, P, k, views[0], views[1], None, o))
        outs.append(o)
    ctx.obmc_batch(jobs)
    for k in range(3):
        assert np.array_equal(outs[k].download(), wants[k]), k


@pytest.mark.parametrize("hs,vs,prec,weights", [(1, 1, 2, (1, 1, 1)), (1, 0, 0, (1, 1, 1)), (1, 1, 3, (3, -1, 1))])
def test_the_have_cuda_branch_renamed(ctx, hs, vs, prec, weights):
    """r06: the reference's HAVE_CUDA branches with `cuda` spelt `hip`, call for call --
    x_render_motion: mc_tmp_frame = S16 frame of the transform's padded size; schro_motion_render_cuda (motion,
    mc_tmp_frame) (schrodecoder.c:1742-1760); x_combine: schro_gpuframe_add (picture->frame, mc_tmp_frame)
    (:1908-1910), schro_gpuframe_convert (planar_output_frame, combined_frame) (:2011).  The picture equals the CPU
    call's fused form (the oracle) wherever the 16-bit sums do not wrap -- they cannot for legal weights --, and the
    s16 frames equal the reference's own Orc kernels on the oracle's accumulator for any weights."""
    w, h, depth, filt = 200, 120, 3, 0
    lib = ctx.lib
    pd = dims(w, h, hs, vs)
    # (schrodecoder.c:1749-1754: the luma size rounded up to the transform depth + the chroma shift)
    il = (round_up(h, depth + vs), round_up(w, depth + hs))
    iw = [il, (il[0] >> vs, il[1] >> hs), (il[0] >> vs, il[1] >> hs)]
    P = synth.motion_params(w, h, 12, 8, prec, weights, (hs, vs))
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 24 << prec, seed=9)
    params = frames.make_params(
        wavelet_filter_index=filt, transform_depth=depth, iwt_luma_width=iw[0][1], iwt_luma_height=iw[0][0],
        iwt_chroma_width=iw[1][1], iwt_chroma_height=iw[1][0], num_refs=2, xblen_luma=12, yblen_luma=12,
        xbsep_luma=8, ybsep_luma=8, mv_precision=prec, picture_weight_bits=weights[2], picture_weight_1=weights[0],
        picture_weight_2=weights[1], x_num_blocks=P["x_num_blocks"], y_num_blocks=P["y_num_blocks"])
    resid = [synth.image_s(ih, iwd, np.int16, seed=60 + k) for k, (ih, iwd) in enumerate(iw)]
    coeffs = [O.forward_iwt(r, depth, filt) for r in resid]
    fmt16, fmt8 = frames.frame_format(np.int16, hs, vs), frames.frame_format(np.uint8, hs, vs)
    frame = frames.DeviceFrame(ctx, fmt16, iw[0][1], iw[0][0])
    sa.check(lib.schro_frame_inverse_iwt_transform_hip(frame.ptr(), frames.HostFrame(coeffs, hs, vs).ptr(), C.byref(params)))
    res_np = frame.download()
    refs_np = [[synth.picture_u8(ph, pw, seed=70 + 10 * r + k) for k, (ph, pw) in enumerate(pd)] for r in range(2)]
    refs = []
    for r in range(2):
        d = frames.DeviceFrame(ctx, fmt8, w, h).upload(frames.HostFrame(refs_np[r], hs, vs))
        if prec > 0:
            u = frames.DeviceFrame(ctx, fmt8, w, h, upsampled=True)
            sa.check(lib.schro_upsampled_hipframe_upsample(u.ptr(), d.ptr()))
            refs.append(u)
        else:
            refs.append(d)
    # x_render_motion, the HAVE_CUDA branch: an S16 mc_tmp_frame of the padded size
    mc_tmp = frames.DeviceFrame(ctx, fmt16, iw[0][1], iw[0][0])
    motion = _lib.Motion(refs[0].ptr(), refs[1].ptr(), mv.ctypes.data, C.pointer(params))
    sa.check(lib.schro_motion_render_hip(C.byref(motion), mc_tmp.ptr(), None, 0, None))
    got_pred = mc_tmp.download()
    accs = []
    for k, (ph, pw) in enumerate(pd):
        _, acc = O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(refs_np[0][k], upsample=prec > 0),
                                 O.UpComp(refs_np[1][k], upsample=prec > 0), np.zeros((ph, pw), np.int16), pw, ph, return_acc=True)
        accs.append(acc)
        assert np.array_equal(got_pred[k][:ph, :pw], O.rrshift6_s16(acc)), k
    # x_combine: schro_gpuframe_add (picture->frame, mc_tmp_frame); schro_gpuframe_convert (output, picture->frame)
    sa.check(lib.schro_hipframe_add(frame.ptr(), mc_tmp.ptr()))
    got_sum = frame.download()
    out = frames.DeviceFrame(ctx, fmt8, w, h)
    sa.check(lib.schro_hipframe_convert(out.ptr(), frame.ptr()))
    got = out.download()
    for k, (ph, pw) in enumerate(pd):
        want_sum = O.frame_add(res_np[k][:ph, :pw], O.rrshift6_s16(accs[k]))
        assert np.array_equal(got_sum[k][:ph, :pw], want_sum), k
        assert np.array_equal(got[k], O.convert_u8(want_sum, pw, ph)), k
        if weights == (1, 1, 1):        # ... which is the CPU call's fused picture
            fused = O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(refs_np[0][k], upsample=prec > 0),
                                    O.UpComp(refs_np[1][k], upsample=prec > 0), res_np[k], pw, ph)
            assert np.array_equal(got[k], fused), k
    # a u8 source: schro_gpuframe_add's other case (schrogpuframe.c:272-283)
    sa.check(lib.schro_hipframe_add(frame.ptr(), refs[0].ptr() if prec == 0 else out.ptr()))
    src8 = refs_np[0] if prec == 0 else got
    for k, (ph, pw) in enumerate(pd):
        assert np.array_equal(frame.download()[k][:ph, :pw], O.frame_add(got_sum[k][:ph, :pw], src8[k])), k
    for f in (frame, out, mc_tmp) + tuple(refs):
        f.unref()
