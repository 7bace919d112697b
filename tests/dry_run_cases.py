"""The library's host code under the sanitizers, WITHOUT a device (run by tests/test_sanitizers.py in a child
process against libschro_hip_dry_asan.so / _dry_tsan.so -- schroedinger_amd/csrc/schro_hip_dry.h: the HIP runtime's
entry points are host stand-ins, kernel launches are dropped).  Nothing is computed and nothing is compared: what
runs is everything in front of and around the launches -- job tables, OBMC tile orders / tile records / weight
tables, the wavelet's level geometry and chain orders, dequantisation plans, the frame layer, the scheduler with real
contexts on several threads -- with the fuzz file's random geometries, so that AddressSanitizer,
UndefinedBehaviorSanitizer and ThreadSanitizer see the index arithmetic a GPU box keeps to itself.

Not collected by a plain `pytest tests/` (the name): the product library has no dry mode."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import schroedinger_amd as sa
import synth
from schroedinger_amd import _lib, frames

if "dry" not in os.path.basename(os.environ.get("SCHRO_HIP_LIB", "")):
    pytest.skip("dry-run cases need SCHRO_HIP_LIB = a libschro_hip_dry_*.so", allow_module_level=True)

SCALE = int(os.environ.get("SCHRO_DRY_SCALE", "1"))


@pytest.fixture(scope="module")
def ctx():
    c = sa.Context(0)
    yield c
    c.close()


def comp_size(w, h, k, chroma):
    return (w, h) if k == 0 else (-(-w // (1 << chroma[0])), -(-h // (1 << chroma[1])))


def obmc_case(ctx, w, h, blen, sep, prec, weights, chroma, pair, pred):
    P = synth.motion_params(w, h, blen, sep, prec, weights, chroma)
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 40 << prec, seed=w + h)
    d_mv = ctx.upload_bytes(mv)
    keep, jobs = [d_mv], []
    pair = pair and prec > 0 and chroma[0] == 1
    pair_hp = [ctx.hp_plane(*comp_size(w, h, 1, chroma)[::-1], pair=True) for _ in range(2)] if pair else None
    for k in range(3):
        cw, ch = comp_size(w, h, k, chroma)
        if prec == 0:
            g1, g2 = ctx.plane(ch, cw, np.uint8), ctx.plane(ch, cw, np.uint8)
        elif pair and k:
            g1, g2 = pair_hp
        else:
            g1, g2 = ctx.hp_plane(ch, cw), ctx.hp_plane(ch, cw)
        out = ctx.plane(ch, cw, np.int16 if pred == 2 else np.uint8)
        res = None if pred else ctx.plane(ch + 8, cw + 16, np.int16)
        jobs.append(sa.obmc_plane(d_mv, P, k, g1, g2, res, out, prediction_only=pred))
        keep += [g1, g2, out] + ([res] if res is not None else [])
    try:
        ctx.obmc_batch(jobs)
    except sa.SchroHipError as e:
        # refusals the API documents (weights a u8 prediction cannot carry) are answers, not findings
        assert "prediction_only" in str(e) or "picture_weight_bits" in str(e), str(e)
    for p in set(keep) | set(pair_hp or []):
        p.free()


def test_obmc_geometries(ctx):
    rng = np.random.default_rng(303)
    seps = [4, 8, 12, 16, 24, 32]
    for rnd in range(300 * SCALE):
        sep = seps[int(rng.integers(0, len(seps)))]
        blen = min(sep + 4 * int(rng.integers(0, sep // 4 + 1)), 2 * sep, 64)
        w, h = int(rng.integers(blen, 520)), int(rng.integers(blen, 280))
        prec = int(rng.integers(0, 4))
        chroma = [(0, 0), (1, 0), (1, 1)][int(rng.integers(0, 3))]
        weights = [(1, 1, 1), (1, 1, 1), (1, 1, 1), (2, 3, 1), (3, 5, 3), (1, 2, 2)][int(rng.integers(0, 6))]
        obmc_case(ctx, w, h, blen, sep, prec, weights, chroma, pair=bool(rng.integers(0, 2)), pred=int(rng.integers(0, 3)))
    # the bench's sizes: every standard block set at 1080p / 2160p, all precisions, both chroma forms, a whole batch
    for (w, h) in ((1920, 1080), (3840, 2160)):
        for (blen, sep) in ((8, 4), (12, 8), (16, 12), (24, 16), (16, 8), (24, 12), (32, 16)):
            for prec in range(4):
                obmc_case(ctx, w, h, blen, sep, prec, (1, 1, 1), (1, 1), pair=True, pred=1)
    obmc_case(ctx, 7680, 4320, 12, 8, 2, (1, 1, 1), (1, 0), pair=True, pred=0)


def test_obmc_batches_of_unlike_pictures(ctx):
    """Eight pictures of one geometry and a few others in ONE call (launch groups, tile orders over several references,
    the weight-table dedupe, job tables near the slot's size)."""
    rng = np.random.default_rng(9)
    for rnd in range(6 * SCALE):
        jobs, keep = [], []
        for pic in range(int(rng.integers(2, 12))):
            w, h = [(640, 360), (352, 288), (1920, 1080)][int(rng.integers(0, 3))]
            blen, sep = [(12, 8), (8, 4), (24, 16), (16, 12)][int(rng.integers(0, 4))]
            prec = int(rng.integers(0, 4))
            P = synth.motion_params(w, h, blen, sep, prec, (1, 1, 1), (1, 1))
            d_mv = ctx.upload_bytes(synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 16 << prec, seed=pic))
            keep.append(d_mv)
            pair = [ctx.hp_plane(h // 2, w // 2, pair=True) for _ in range(2)] if prec else None
            for k in range(3):
                cw, ch = comp_size(w, h, k, (1, 1))
                if prec == 0:
                    g = [ctx.plane(ch, cw, np.uint8) for _ in range(2)]
                elif k:
                    g = pair
                else:
                    g = [ctx.hp_plane(ch, cw) for _ in range(2)]
                out = ctx.plane(ch, cw, np.uint8)
                keep += list(g) + [out]
                jobs.append(sa.obmc_plane(d_mv, P, k, g[0], g[1], None, out, prediction_only=1))
        ctx.obmc_batch(jobs)
        for p in set(keep):
            p.free()


def test_iiwt_geometries(ctx):
    rng = np.random.default_rng(101)
    for rnd in range(200 * SCALE):
        filt, depth = int(rng.integers(0, 7)), int(rng.integers(1, 5))
        dtype = [np.int16, np.int32][int(rng.integers(0, 2))]
        pairs, keep = [], []
        for _ in range(int(rng.integers(1, 6))):
            unit = 1 << depth
            h, w = unit * int(rng.integers(1, 80)), unit * int(rng.integers(1, 140))
            src = ctx.plane(h, w + int(rng.integers(0, 3)) * 8, dtype)
            src.width = w
            combine = int(rng.integers(0, 3))
            if combine == 0:
                dst = ctx.plane(h, w, dtype)
                pairs.append((src, dst))
                keep += [src, dst]
            else:
                oh, ow = h - int(rng.integers(0, unit)), w - int(rng.integers(0, unit))
                oh, ow = max(oh, 1), max(ow, 1)
                out = ctx.plane(oh, ow, np.uint8)
                pred = ctx.plane(oh, ow, np.uint8) if combine == 1 else None
                pairs.append((src, out, pred))
                keep += [src, out] + ([pred] if pred is not None else [])
        if len({len(p) for p in pairs}) == 1:          # (a call's planes are all of one form)
            ctx.iiwt_batch(pairs, depth, filt)
        for p in keep:
            p.free()
    # the headline's batch: 24 planes of 2160p 4:2:0, combine form, and 1080p with its padded chroma
    for (w, h) in ((3840, 2160), (1920, 1080)):
        pairs, keep = [], []
        for pic in range(8):
            for (ph, pw) in ((h, w), (h // 2, w // 2), (h // 2, w // 2)):
                ih, iw = -(-ph // 8) * 8, -(-pw // 8) * 8
                t = (ctx.plane(ih, iw, np.int16), ctx.plane(ph, pw, np.uint8), ctx.plane(ph, pw, np.uint8))
                pairs.append(t)
                keep += list(t)
        ctx.iiwt_batch(pairs, 3, 0)
        for p in keep:
            p.free()


def test_frameops_geometries(ctx):
    rng = np.random.default_rng(202)
    for rnd in range(120 * SCALE):
        h, w = int(rng.integers(1, 300)), int(rng.integers(1, 800))
        src, dst = ctx.plane(h, w, np.uint8), ctx.hp_plane(h, w)
        ctx.upsample_batch([(src, dst)])
        su, sv, dp = ctx.plane(h, w, np.uint8), ctx.plane(h, w, np.uint8), ctx.hp_plane(h, w, pair=True)
        ctx.upsample_batch([((su, sv), dp)])
        dtype = [np.int16, np.int32][rnd & 1]
        res, out = ctx.plane(h + int(rng.integers(0, 9)), w + int(rng.integers(0, 17)), dtype), ctx.plane(h, w, np.uint8)
        ctx.convert_u8_batch([(res, out)])
        d16 = ctx.plane(h, w, np.int16)
        ctx.add_batch([(d16, out)])
        ctx.add_batch([(d16, ctx.plane(h + 1, w + 2, np.int16))])
        ctx.shift_right_batch([d16], int(rng.integers(0, 8)))
        for p in (src, dst, su, sv, dp, res, out, d16):
            p.free()


def test_dequant_and_lowdelay_geometries(ctx):
    rng = np.random.default_rng(404)
    for rnd in range(40 * SCALE):
        depth = int(rng.integers(1, 5))
        unit = 1 << depth
        w, h = unit * int(rng.integers(2, 60)), unit * int(rng.integers(2, 40))
        dtype = [np.int16, np.int32][rnd & 1]
        plane = ctx.plane(h, w, dtype)
        hc = [int(rng.integers(1, 5)) for _ in range(depth + 1)]
        vc = [int(rng.integers(1, 4)) for _ in range(depth + 1)]
        cbs = ctx.codeblock_layout(w, h, depth, hc, vc, plane.stride, plane.dtype.itemsize)
        off = 0
        for cb in cbs:
            n = cb.width * cb.height
            if rng.integers(0, 3) == 0 or n == 0:
                cb.src_offset = -1
            else:
                cb.src_bytes = 2
                cb.src_offset = off
                off += 2 * n
            cb.quant_index = int(rng.integers(0, 61))
        blob = ctx.upload_bytes(np.zeros(max(off, 2), np.uint8))
        jobs = [(plane, blob, cbs, bool(rnd & 2))]
        ctx.dequant_batch(jobs, arith=0)
        plan = ctx.dequant_plan(jobs, arith=0)
        plan.run(jobs)
        plan.free()
        plane.free()
        blob.free()


def test_frame_layer_stage_calls(ctx):
    """x_wavelet_transform, x_upsample, x_render_motion (all three forms), x_combine through the SchroFrame-shaped calls."""
    lib = ctx.lib
    for (w, h, hs, vs, prec, blk) in ((320, 240, 1, 1, 2, (12, 8)), (200, 120, 1, 0, 0, (8, 4)), (208, 112, 0, 0, 3, (16, 12)),
                                      (352, 288, 1, 1, 0, (24, 16))):
        depth, filt = 3, 0
        il = ((h + (1 << depth + vs) - 1) >> (depth + vs) << (depth + vs), (w + (1 << depth + hs) - 1) >> (depth + hs) << (depth + hs))
        iw = [il, (il[0] >> vs, il[1] >> hs), (il[0] >> vs, il[1] >> hs)]
        P = synth.motion_params(w, h, blk[0], blk[1], prec, (1, 1, 1), (hs, vs))
        mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 24 << prec, seed=9)
        params = frames.make_params(
            wavelet_filter_index=filt, transform_depth=depth, iwt_luma_width=iw[0][1], iwt_luma_height=iw[0][0],
            iwt_chroma_width=iw[1][1], iwt_chroma_height=iw[1][0], num_refs=2, xblen_luma=blk[0], yblen_luma=blk[0],
            xbsep_luma=blk[1], ybsep_luma=blk[1], mv_precision=prec, picture_weight_bits=1, picture_weight_1=1,
            picture_weight_2=1, x_num_blocks=P["x_num_blocks"], y_num_blocks=P["y_num_blocks"])
        fmt16, fmt8 = frames.frame_format(np.int16, hs, vs), frames.frame_format(np.uint8, hs, vs)
        coeffs = [np.zeros(s, np.int16) for s in iw]
        frame = frames.DeviceFrame(ctx, fmt16, iw[0][1], iw[0][0])
        sa.check(lib.schro_frame_inverse_iwt_transform_hip(frame.ptr(), frames.HostFrame(coeffs, hs, vs).ptr(), C.byref(params)))
        refs = []
        for r in range(2):
            d = frames.DeviceFrame(ctx, fmt8, w, h)
            if prec > 0:
                u = frames.DeviceFrame(ctx, fmt8, w, h, upsampled=True)
                sa.check(lib.schro_upsampled_hipframe_upsample(u.ptr(), d.ptr()))
                refs.append(u)
                d.unref()
            else:
                refs.append(d)
        motion = _lib.Motion(refs[0].ptr(), refs[1].ptr(), mv.ctypes.data, C.pointer(params))
        out, mc8, mc16 = (frames.DeviceFrame(ctx, fmt8, w, h), frames.DeviceFrame(ctx, fmt8, w, h),
                          frames.DeviceFrame(ctx, fmt16, iw[0][1], iw[0][0]))
        sa.check(lib.schro_motion_render_hip(C.byref(motion), None, frame.ptr(), 1, out.ptr()))
        sa.check(lib.schro_motion_render_hip(C.byref(motion), mc8.ptr(), None, 0, None))
        sa.check(lib.schro_motion_render_hip(C.byref(motion), mc16.ptr(), None, 0, None))
        sa.check(lib.schro_hipframe_add(frame.ptr(), mc16.ptr()))
        sa.check(lib.schro_hipframe_add(frame.ptr(), mc8.ptr()))
        sa.check(lib.schro_hipframe_convert(out.ptr(), frame.ptr()))
        sa.check(lib.schro_frame_inverse_iwt_transform_combine_hip(out.ptr(), frames.HostFrame(coeffs, hs, vs).ptr(), C.byref(params), mc8.ptr()))
        host = out.download()
        assert host[0].shape == (h, w)
        for f in (frame, out, mc8, mc16) + tuple(refs):
            f.unref()


def test_contexts_on_several_threads(ctx):
    """Contexts of their own on four threads at once (the exec-domain threads of a multi-GPU decoder) plus the shared
    error state: what ThreadSanitizer is here for."""
    errs = []

    def worker(n):
        try:
            c = sa.Context(n % 8)
            for rnd in range(10):
                obmc_case(c, 160 + 16 * n, 96, 12, 8, rnd % 4, (1, 1, 1), (1, 1), pair=True, pred=1)
                src, dst = c.plane(64, 96, np.uint8), c.hp_plane(64, 96)
                c.upsample_batch([(src, dst)])
                c.synchronize()
            c.close()
        except Exception as e:          # noqa: BLE001 -- reported to the main thread
            errs.append(repr(e))
    ths = [threading.Thread(target=worker, args=(n,)) for n in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs


def test_scheduler_with_contexts_and_moving_references():
    """The scheduler on three (dry) devices with REAL contexts: reference pictures allocate and publish an upsampled
    frame, dependents of other chains fetch it (a peer copy behind the owner's event), pictures retire while others
    run: the records' lifetimes and the per-device threads under ThreadSanitizer / AddressSanitizer."""
    sched = sa.Scheduler(3)
    assert sched.n_devices == 3 and all(c is not None for c in sched.contexts)
    fmt8 = frames.frame_format(np.uint8, 1, 1)
    w, h = 176, 144
    P = synth.motion_params(w, h, 12, 8, 2, (1, 1, 1), (1, 1))
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 32, seed=3)
    keep, lock = {}, threading.Lock()

    def reference(number):
        def run(ctx, dev):
            plain = frames.DeviceFrame(ctx, fmt8, w, h)
            up = frames.DeviceFrame(ctx, fmt8, w, h, upsampled=True)
            sa.check(ctx.lib.schro_upsampled_hipframe_upsample(up.ptr(), plain.ptr()))
            sched.publish_reference(dev, up.ptr())
            with lock:
                keep[number] = (plain, up)
            return 0
        return run

    def dependent(number, refs):
        def run(ctx, dev):
            f = [C.cast(sched.reference_frame(dev, n), C.POINTER(_lib.Frame)) for n in refs]
            assert all(f)
            d_mv = ctx.upload_bytes(mv)
            planes, outs = [], []
            for k in range(3):
                class V:
                    pass
                views = []
                for fr in f:
                    v = V()
                    v.ptr, v.stride = fr.contents.components[k].data, fr.contents.components[k].stride
                    v.pair = k > 0 and fr.contents.is_upsampled == 2
                    views.append(v)
                out = ctx.plane(h >> (k > 0), w >> (k > 0), np.uint8)
                outs.append(out)
                planes.append(sa.obmc_plane(d_mv, P, k, views[0], views[-1], None, out, prediction_only=1))
            ctx.obmc_batch(planes)
            ctx.synchronize()
            for p in outs + [d_mv]:
                p.free()
            return 0
        return run

    # three chains of anchors; B pictures predict from anchors of two different chains (a foreign reference moves)
    anchors = []
    for n in range(9):
        refs = [anchors[-3]] if len(anchors) >= 3 else []
        sched.submit(n, refs, True, reference(n) if not refs else reference(n))
        anchors.append(n)
    for n in range(9, 27):
        a, b = anchors[(n * 5) % 9], anchors[(n * 7 + 1) % 9]
        sched.submit(n, [a, b], False, dependent(n, [a, b]))
        if n % 4 == 0:
            sched.retire(n)
    assert sched.wait() == 0
    assert sched.moves() > 0
    for n in range(9):
        sched.retire(n)
    sched.close()
    for plain, up in keep.values():
        pass                                            # (the scheduler released the published frames; the plain ones go with their contexts)
