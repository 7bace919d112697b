"""GPU parity: core-syntax dequantisation on the device (schro_hip_dequant_batch)
against oracle/oracle_dequant.c -- which is pinned on the reference's compiled
orc_dequantise_* kernels -- and, end to end, the reference's own stream decoded from its
QUANTISED values: dequantise + intra DC prediction + inverse wavelet + OBMC / convert on the
device, every one of the 100 pictures equal to the oracle's and carrying its digest."""
import json
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
import schroedinger_amd as sa
import stream_lib as S

sys.path.insert(0, os.path.join(S.ROOT, "oracle"))
import dirac_stream as D  # noqa: E402

pytestmark = pytest.mark.gpu


def pack_codeblocks(plane_shape, itemsize, stride, depth, quant, records):
    """(quantised plane, front-end records) -> (values blob, SchroHipCodeblock tuples): what a
    host decoder hands over.  Values of non-zero codeblocks only, 1 / 2 / 4 bytes each."""
    probe = np.zeros(plane_shape, np.int8)
    blob, cbs = bytearray(), []
    for (index, x0, y0, x1, y1, zero, qi) in records:
        band = D.subband_view(probe, depth, index)
        # byte offset / stride of the sub-band inside the device plane (same geometry, device stride)
        off_elems = (band.__array_interface__["data"][0] - probe.__array_interface__["data"][0])
        row0, col0 = divmod(off_elems, probe.strides[0])
        lvl_stride = band.strides[0] // probe.strides[0]
        dst_off = (row0 + y0 * lvl_stride) * stride + (col0 + x0) * itemsize
        dst_stride = lvl_stride * stride
        if zero:
            cbs.append((dst_off, dst_stride, x1 - x0, y1 - y0, -1, 0, qi))
            continue
        q = D.subband_view(quant, depth, index)[y0:y1, x0:x1]
        m = int(np.abs(q).max()) if q.size else 0
        dt = np.int8 if m < 128 else (np.int16 if m < 32768 else np.int32)
        while len(blob) % np.dtype(dt).itemsize:
            blob.append(0)
        cbs.append((dst_off, dst_stride, x1 - x0, y1 - y0, len(blob), np.dtype(dt).itemsize, qi))
        blob += np.ascontiguousarray(q, dtype=dt).tobytes()
    return np.frombuffer(bytes(blob) or b"\0", np.uint8), cbs


def synthetic_records(h, w, depth, rng, zero_share=0.4, counts=None):
    """A codeblock partition of every sub-band (schrodecoder.c:3572-3596 geometry) with random
    zero flags and quantiser indices; counts: (horizontal, vertical) codeblocks per sub-band, else random."""
    records = []
    probe = np.zeros((h, w), np.int8)
    for index in range(1 + 3 * depth):
        band = D.subband_view(probe, depth, index)
        bh, bw = band.shape
        ncx, ncy = counts if counts else (int(rng.integers(1, 6)), int(rng.integers(1, 5)))
        cbw = bw // ncx
        inc = bw - ncx * cbw
        for cy in range(ncy):
            y0, y1 = (bh * cy) // ncy, (bh * (cy + 1)) // ncy
            xmin = acc = 0
            for cx in range(ncx):
                x0 = xmin
                xmin += cbw
                acc += inc
                if acc >= ncx:
                    acc -= ncx
                    xmin += 1
                records.append((index, x0, y0, xmin, y1, bool(rng.random() < zero_share), int(rng.integers(0, 61))))
    return records


@pytest.mark.parametrize("dtype,arith", [(np.int16, 0), (np.int16, 1), (np.int32, 0)])
def test_synthetic_codeblocks(ctx, dtype, arith):
    rng = np.random.default_rng(11)
    jobs, want, outs = [], [], []
    for (h, w, depth, span, intra) in [(64, 96, 2, 100, 1), (144, 176, 3, 3000, 0), (48, 40, 1, 40000, 0),
                                       (270 * 4, 480 * 4, 4, 90, 1)]:
        quant = rng.integers(-span, span + 1, (h, w)).astype(np.int32)
        quant[rng.random((h, w)) < 0.5] = 0
        records = synthetic_records(h, w, depth, rng)
        dst = ctx.plane(h, w, dtype).fill(0x5a)
        blob, cbs = pack_codeblocks((h, w), np.dtype(dtype).itemsize, dst.stride, depth, quant, records)
        jobs.append((dst, ctx.upload_bytes(blob), cbs, intra))
        ref = np.full((h, w), 0x5a5a5a5a & (0xffff if dtype == np.int16 else 0xffffffff), np.uint32).astype(dtype)
        for (index, x0, y0, x1, y1, zero, qi) in records:
            band, qb = D.subband_view(ref, depth, index), D.subband_view(quant, depth, index)
            if x1 > x0 and y1 > y0:
                O.dequant_codeblock(band[y0:y1, x0:x1], None if zero else qb[y0:y1, x0:x1], qi, intra, arith)
        want.append(ref)
        outs.append(dst)
    ctx.dequant_batch(jobs, arith)
    for n, (dst, ref) in enumerate(zip(outs, want)):
        assert np.array_equal(dst.download(), ref), n


def test_more_codeblocks_than_a_table_slot_holds(ctx):
    # 5200 codeblocks in one plane + 900 in another: the job table goes up through the large-table
    # path (beyond 64 KB) and the kernel finds a tile's codeblock with all three probes (> 4096 jobs);
    # codeblocks down to 1 x 1 sample
    rng = np.random.default_rng(23)
    jobs, want, outs = [], [], []
    for (h, w, depth, counts) in [(512, 512, 4, (20, 20)), (96, 160, 2, (10, 9))]:
        quant = rng.integers(-300, 301, (h, w)).astype(np.int32)
        quant[rng.random((h, w)) < 0.5] = 0
        records = synthetic_records(h, w, depth, rng, counts=counts)
        dst = ctx.plane(h, w, np.int16).fill(0x5a)
        blob, cbs = pack_codeblocks((h, w), 2, dst.stride, depth, quant, records)
        jobs.append((dst, ctx.upload_bytes(blob), cbs, 0))
        ref = np.full((h, w), 0x5a5a, np.uint16).astype(np.int16)
        for (index, x0, y0, x1, y1, zero, qi) in records:
            band, qb = D.subband_view(ref, depth, index), D.subband_view(quant, depth, index)
            if x1 > x0 and y1 > y0:
                O.dequant_codeblock(band[y0:y1, x0:x1], None if zero else qb[y0:y1, x0:x1], qi, 0, 0)
        want.append(ref)
        outs.append(dst)
    assert sum(len(c) for _, _, c, _ in jobs) > 4096
    for rnd in range(3):            # (the large tables take turns: every one of them gets used)
        for dst in outs:
            dst.fill(0x5a)
        ctx.dequant_batch(jobs, 0)
        for n, (dst, ref) in enumerate(zip(outs, want)):
            assert np.array_equal(dst.download(), ref), (n, rnd)


def test_bad_arguments_are_refused(ctx):
    dst = ctx.plane(16, 16, np.int16)
    blob = ctx.upload_bytes(np.zeros(512, np.uint8))
    with pytest.raises(sa.SchroHipError):       # 3-byte values
        ctx.dequant_batch([(dst, blob, [(0, dst.stride, 4, 4, 0, 3, 10)], 0)])
    with pytest.raises(sa.SchroHipError):       # quantiser index beyond the tables
        ctx.dequant_batch([(dst, blob, [(0, dst.stride, 4, 4, 0, 1, 61)], 0)])
    with pytest.raises(sa.SchroHipError):       # the 16-bit arithmetic on an s32 frame
        ctx.dequant_batch([(ctx.plane(16, 16, np.int32), blob, [(0, 64, 4, 4, 0, 1, 10)], 0)], arith=1)


def test_whole_stream_from_quantised_values(ctx):
    md5 = json.load(open(os.path.join(S.GOLDEN, "stream_md5.json")))["oracle"]
    decoded, count, sent, dense = {}, 0, 0, 0
    for rec in S.decode_stream(S.load_stream(), S.load_tables(), quantised=True):
        number, refs, want = rec["number"], rec["refs"], rec["out"]
        intra = rec["num_refs"] == 0
        out = [ctx.plane(w.shape[0], w.shape[1], np.uint8).fill(0x11) for w in want]
        tmp = []
        if rec["zero_residual"]:
            res = [ctx.upload(np.zeros(w.shape, np.int16)) for w in want]
        else:
            depth = rec["depth"]
            co, jobs = [], []
            for k in range(3):
                q = rec["quant"][k]
                d = ctx.plane(q.shape[0], q.shape[1], np.int16).fill(0x5a)
                blob, cbs = pack_codeblocks(q.shape, 2, d.stride, depth, q, rec["codeblocks"][k])
                sent += blob.size + 24 * len(cbs)
                dense += q.size * 2
                vals = ctx.upload_bytes(blob)
                jobs.append((d, vals, cbs, intra))
                co.append(d)
                tmp.append(vals)
            ctx.dequant_batch(jobs, 0)
            if intra:       # schrodecoder.c:3629-3636: the LL band of an intra picture
                ctx.dc_predict_batch([sa.SubPlane(c, 0, 0, c.height >> depth, c.width >> depth, c.stride << depth)
                                      for c in co])
            for k in range(3):
                assert np.array_equal(co[k].download(), rec["coeffs"][k]), "picture %d coefficients %d" % (number, k)
            res = [ctx.plane(c.height, c.width, np.int16) for c in co]
            ctx.iiwt_batch(list(zip(co, res)), depth, rec["wavelet"])
            tmp += co
        if intra:
            ctx.convert_u8_batch(list(zip(res, out)))
        else:
            d_mv = ctx.upload_bytes(rec["mv"])
            ctx.obmc_batch([sa.obmc_plane(d_mv, rec["params"], k, decoded[refs[0]][k],
                                          decoded[refs[1] if len(refs) > 1 else refs[0]][k], res[k], out[k])
                            for k in range(3)])
        got = [o.download() for o in out]
        for k in range(3):
            assert np.array_equal(got[k], want[k]), "picture %d component %d" % (number, k)
        assert D.frame_md5(got) == md5[number]
        for p in tmp + res:
            p.free()
        if rec["is_ref"]:
            decoded[number] = out
        else:
            for p in out:
                p.free()
        count += 1
    assert count == 100
    # what crossed the host boundary instead of dense s16 coefficient frames
    print("quantised hand-over: %.1f %% of the dense coefficient bytes" % (100.0 * sent / dense))
    assert sent < dense / 2


def test_a_plan_reused_over_different_pictures(ctx):
    """r04: schro_hip_dequant_plan_* -- the codeblock GEOMETRY of a picture format lives on the device; a run
    uploads the decoder's records as they are and the device reads src_offset / src_bytes / quant_index itself.
    One plan, four pictures with different zero codeblocks, value widths, quantisers and values: each equals the
    oracle (and schro_hip_dequant_batch); records whose geometry is not the plan's are refused."""
    w, h, depth = 360, 208, 3
    hc, vc = [1, 3, 4, 5], [1, 2, 3, 4]
    for dtype, arith in ((np.int16, 0), (np.int16, 1), (np.int32, 0)):
        itemsize = np.dtype(dtype).itemsize
        planes = [ctx.plane(h, w, dtype).fill(0x5b) for _ in range(2)]
        layout = ctx.codeblock_layout(w, h, depth, hc, vc, planes[0].stride, itemsize)
        geo = [(t.dst_offset, t.dst_stride, t.width, t.height) for t in layout]

        def picture(seed):
            rng = np.random.default_rng(seed)
            jobs, wants, keep = [], [], []
            for pl, intra in zip(planes, (True, False)):
                blob, cbs, off = [], [], 0
                want = np.zeros((h, w), dtype)
                for (dst_off, stride, cw, chh) in geo:
                    if cw == 0 or chh == 0 or rng.random() < 0.3:
                        cbs.append((dst_off, stride, cw, chh, -1, 0, int(rng.integers(0, 61))))
                        continue
                    sb = int(rng.choice([1, 2, 4]))
                    qi = int(rng.integers(0, 40))
                    lim = 100 if sb == 1 else 3000
                    q = rng.integers(-lim, lim + 1, (chh, cw)).astype({1: np.int8, 2: np.int16, 4: np.int32}[sb])
                    off = (off + 3) // 4 * 4
                    cbs.append((dst_off, stride, cw, chh, off, sb, qi))
                    blob.append((off, q))
                    off += q.nbytes
                    y0, xb = divmod(dst_off, pl.stride)
                    step = stride // pl.stride          # the sub-band's rows lie 2^level frame rows apart
                    O.dequant_codeblock(want[y0:y0 + chh * step:step, xb // itemsize:xb // itemsize + cw], q, qi, intra, arith)
                raw = np.zeros((1, max(off, 4)), np.uint8)
                for o, q in blob:
                    raw[0, o:o + q.nbytes] = q.view(np.uint8).reshape(-1)
                dev = ctx.upload(raw)
                keep.append(dev)
                jobs.append((pl, dev, cbs, intra))
                wants.append(want)
            return jobs, wants, keep
        first, _, keep0 = picture(1)
        plan = ctx.dequant_plan(first, arith)
        for seed in (1, 2, 3, 4):
            jobs, wants, keep = picture(seed)
            for pl in planes:
                pl.fill(0x5b)
            plan.run(jobs)
            got = [pl.download() for pl in planes]
            for pl in planes:
                pl.fill(0x3c)
            ctx.dequant_batch(jobs, arith)
            for k, pl in enumerate(planes):
                assert np.array_equal(got[k], pl.download()), (dtype, arith, seed, k)
            # (the oracle wrote only where codeblocks lie: compare there -- everything, the planes are covered)
            for k in range(2):
                assert np.array_equal(got[k], wants[k]), (dtype, arith, seed, k)
            for d in keep:
                d.free()
        # a record with another geometry, a wild quantiser
        jobs, _, keep = picture(9)
        bad = list(jobs[0][2])
        bad[3] = (bad[3][0], bad[3][1], bad[3][2] + 1) + tuple(bad[3][3:])
        with pytest.raises(sa.SchroHipError, match="differs from the plan"):
            plan.run([(jobs[0][0], jobs[0][1], bad, True), jobs[1]])
        plan.free()
        for p in planes + keep + keep0:
            p.free()


@pytest.mark.parametrize("dtype,noarith,num_refs,on_device", [(np.int16, 0, 0, 0), (np.int16, 1, 1, 1), (np.int32, 0, 0, 1),
                                                            (np.int16, 0, 2, 0), (np.int32, 1, 0, 0)])
def test_the_frame_layer_takes_codeblock_records_and_values(ctx, dtype, noarith, num_refs, on_device):
    """r05: schro_hipframe_dequantise (transform_frame, quantised, params) -- what a patched schro_decoder_decode_subband
    (schrodecoder.c:3525-3640) leaves behind for a picture: zero codeblocks zero-filled, the others dequantised with the
    decoder's arithmetic (the 16-bit Orc program only for is_noarith on an s16 frame), the LL bands of a picture without
    references DC-predicted (:3629-3636).  Values from host memory (staged by the call) and from the device; pictures of two
    geometries after one another (the context's plan is rebuilt, then kept)."""
    import ctypes as C
    from schroedinger_amd import _lib, frames
    lib = ctx.lib
    rng = np.random.default_rng(5 + num_refs)
    intra = num_refs == 0
    arith = 1 if noarith and dtype == np.int16 else 0
    for (iw, ih, depth) in [(96, 64, 2), (96, 64, 2), (352, 288, 3)]:
        dims = [(ih, iw), (ih // 2, iw // 2), (ih // 2, iw // 2)]
        tf = frames.DeviceFrame(ctx, frames.frame_format(dtype, 1, 1), iw, ih)
        # the frame starts from a pattern: every sample is written
        tf.upload(frames.HostFrame([np.full(d, 0x1234, dtype) for d in dims], 1, 1))
        params = frames.make_params(is_noarith=noarith, transform_depth=depth, num_refs=num_refs, iwt_luma_width=iw,
                                    iwt_luma_height=ih, iwt_chroma_width=iw // 2, iwt_chroma_height=ih // 2)
        qp = _lib.QuantisedPicture()
        keep, want = [], []
        for k, (h, w) in enumerate(dims):
            quant = rng.integers(-300, 301, (h, w)).astype(np.int32)
            quant[rng.random((h, w)) < 0.6] = 0
            records = synthetic_records(h, w, depth, rng)
            blob, cbs = pack_codeblocks((h, w), np.dtype(dtype).itemsize, tf.c.components[k].stride, depth, quant, records)
            tab = sa.Context.codeblock_table(cbs)
            qp.codeblocks[k] = C.cast(tab, C.POINTER(_lib.Codeblock))
            qp.ncodeblocks[k] = len(cbs)
            if on_device:
                dev = ctx.upload_bytes(blob)
                qp.values[k] = dev.ptr
                keep.append(dev)
            else:
                qp.values[k] = blob.ctypes.data
            qp.values_bytes[k] = blob.size
            keep += [tab, blob]
            ref = np.zeros((h, w), dtype)
            for (index, x0, y0, x1, y1, zero, qi) in records:
                band, qb = D.subband_view(ref, depth, index), D.subband_view(quant, depth, index)
                if x1 > x0 and y1 > y0:
                    O.dequant_codeblock(band[y0:y1, x0:x1], None if zero else qb[y0:y1, x0:x1], qi, 1 if intra else 0, arith)
            if intra:
                ll = D.subband_view(ref, depth, 0)
                ll[...] = O.dc_predict(ll)
            want.append(ref)
        qp.values_on_device = on_device
        sa.check(lib.schro_hipframe_dequantise(tf.ptr(), C.byref(qp), C.byref(params)))
        got = tf.download()
        for k in range(3):
            assert np.array_equal(got[k], want[k]), (iw, ih, k)
        tf.unref()
        for d in keep:
            if hasattr(d, "free"):
                d.free()


def test_the_frame_layer_refuses_what_is_not_a_quantised_picture(ctx):
    import ctypes as C
    from schroedinger_amd import _lib, frames
    lib = ctx.lib
    params = frames.make_params(transform_depth=1, num_refs=1, iwt_luma_width=32, iwt_luma_height=32, iwt_chroma_width=16,
                                iwt_chroma_height=16)
    tf = frames.DeviceFrame(ctx, frames.frame_format(np.int16, 1, 1), 32, 32)
    qp = _lib.QuantisedPicture()
    assert lib.schro_hipframe_dequantise(tf.ptr(), C.byref(qp), C.byref(params)) == -1       # no records
    assert b"codeblock" in lib.schro_hip_last_error()
    u8 = frames.DeviceFrame(ctx, frames.frame_format(np.uint8, 1, 1), 32, 32)
    assert lib.schro_hipframe_dequantise(u8.ptr(), C.byref(qp), C.byref(params)) == -1       # a u8 frame
    assert lib.schro_hipframe_dequantise(None, C.byref(qp), C.byref(params)) == -1
    tf.unref()
    u8.unref()
