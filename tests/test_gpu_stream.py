"""GPU parity on a real stream (BASELINE config 2): the reference's testsuite/test_stream.drc.
Coefficients and motion vectors of its first pictures (tests/golden/stream_pictures.npz, decoded
from the stream by oracle/dirac_stream.py) go through the HIP path -- inverse wavelet, intra
convert, OBMC from the pictures the GPU itself decoded before -- and every picture must equal
the oracle's, whose first three frames are bit-identical to the reference decoder's
(schro_frame_md5 recorded in SURVEY.md 8(c)); the digests of the GPU's frames are checked too."""
import json
import os
import sys

import numpy as np
import pytest

import schroedinger_amd as sa
import stream_lib as S

sys.path.insert(0, os.path.join(S.ROOT, "oracle"))
import dirac_stream as D  # noqa: E402

pytestmark = pytest.mark.gpu


def test_stream_pictures_through_the_gpu(ctx):
    z = np.load(os.path.join(S.GOLDEN, "stream_pictures.npz"))
    md5 = json.load(open(os.path.join(S.GOLDEN, "stream_md5.json")))
    decoded = {}                                         # picture number -> device u8 planes
    digests = {}
    for n in range(8):
        tag = "p%d_" % n
        number, num_refs, is_ref, zero_residual = [int(v) for v in z[tag + "number"][:4]]
        refs = [int(v) for v in z[tag + "number"][4:]]
        assert not zero_residual
        depth, wavelet = [int(v) for v in z[tag + "transform"]]
        want = [z[tag + "out%d" % k] for k in range(3)]
        co = [ctx.upload(z[tag + "coeff%d" % k]) for k in range(3)]
        res = [ctx.plane(c.height, c.width, np.int16) for c in co]
        ctx.iiwt_batch(list(zip(co, res)), depth, wavelet)
        out = [ctx.plane(w.shape[0], w.shape[1], np.uint8).fill(0x11) for w in want]
        if num_refs == 0:
            ctx.convert_u8_batch(list(zip(res, out)))
        else:
            P = dict(zip(S.PARAM_KEYS, [int(v) for v in z[tag + "params"]]))
            assert P["mv_precision"] == 0                # the stream is full-pel
            d_mv = ctx.upload_bytes(z[tag + "mv"])
            jobs = []
            for k in range(3):
                r1 = decoded[refs[0]][k]
                r2 = decoded[refs[1]][k] if num_refs > 1 else r1
                jobs.append(sa.obmc_plane(d_mv, P, k, r1, r2, res[k], out[k]))
            ctx.obmc_batch(jobs)
        got = [o.download() for o in out]
        for k in range(3):
            assert np.array_equal(got[k], want[k]), "coded picture %d (number %d) component %d" % (n, number, k)
        digests[number] = D.frame_md5(got)
        decoded[number] = out                            # later pictures predict from the GPU's own output
        for p in co + res:
            p.free()
    # the GPU's frames 0, 1, 2 carry the reference decoder's digests
    assert [digests[k] for k in (0, 1, 2)] == md5["reference"]
    for number, d in digests.items():
        assert d == md5["oracle"][number]


def test_whole_stream_through_the_gpu(ctx):
    # all 100 pictures (4 intra-started chains, P and B pictures): bitstream decoded by the oracle's
    # front end, every pixel stage on the GPU, references = the GPU's own earlier output
    md5 = json.load(open(os.path.join(S.GOLDEN, "stream_md5.json")))["oracle"]
    decoded, count = {}, 0
    for rec in S.decode_stream(S.load_stream(), S.load_tables()):
        number, refs = rec["number"], rec["refs"]
        want = rec["out"]
        out = [ctx.plane(w.shape[0], w.shape[1], np.uint8).fill(0x11) for w in want]
        if rec["zero_residual"]:
            res = [ctx.upload(np.zeros(w.shape, np.int16)) for w in want]
            co = []
        else:
            co = [ctx.upload(c) for c in rec["coeffs"]]
            res = [ctx.plane(c.height, c.width, np.int16) for c in co]
            ctx.iiwt_batch(list(zip(co, res)), rec["depth"], rec["wavelet"])
        if rec["num_refs"] == 0:
            ctx.convert_u8_batch(list(zip(res, out)))
        else:
            d_mv = ctx.upload_bytes(rec["mv"])
            jobs = [sa.obmc_plane(d_mv, rec["params"], k, decoded[refs[0]][k],
                                  decoded[refs[1] if len(refs) > 1 else refs[0]][k], res[k], out[k]) for k in range(3)]
            ctx.obmc_batch(jobs)
        got = [o.download() for o in out]
        for k in range(3):
            assert np.array_equal(got[k], want[k]), "picture %d component %d" % (number, k)
        assert D.frame_md5(got) == md5[number]
        for p in co + res:
            p.free()
        if rec["is_ref"]:
            decoded[number] = out
        else:
            for p in out:
                p.free()
        count += 1
    assert count == 100


def test_scaled_vectors_through_the_sub_pel_kernels(ctx):
    # as tests/test_oracle_stream.py::test_scaled_vectors_through_the_sub_pel_path: the stream's
    # blocks with mv_precision p and vectors << p, through the GPU's half-pel upsample and the
    # half- / quarter- / eighth-pel OBMC kernels, must reproduce the full-pel picture
    import oracle_lib as O
    z = np.load(os.path.join(S.GOLDEN, "stream_pictures.npz"))
    out = {int(z["p%d_number" % n][0]): [z["p%d_out%d" % (n, k)] for k in range(3)] for n in range(8)}
    for n in (1, 2, 5):
        tag = "p%d_" % n
        refs = [int(v) for v in z[tag + "number"][4:]]
        depth, wavelet = [int(v) for v in z[tag + "transform"]]
        P = dict(zip(S.PARAM_KEYS, [int(v) for v in z[tag + "params"]]))
        res_np = [O.inverse_iwt(z[tag + "coeff%d" % k], depth, wavelet) for k in range(3)]
        base = z[tag + "mv"].copy()
        vec = (base["flags"] & 3) != 0
        base["v"][vec, 0:2] &= ~1                        # chroma vectors = luma >> 1 (4:2:2)
        want = []
        for k in range(3):
            u = [O.UpComp(out[r][k], upsample=False) for r in refs]
            shape = out[refs[0]][k].shape
            want.append(O.motion_render(base, O.MotionParams(**P), k, u[0], u[1] if len(u) > 1 else None,
                                        res_np[k], shape[1], shape[0]))
        res = [ctx.upload(r) for r in res_np]
        plain = {r: [ctx.upload(out[r][k]) for k in range(3)] for r in refs}
        hp = {r: [ctx.hp_plane(*out[r][k].shape) for k in range(3)] for r in refs}
        ctx.upsample_batch([(plain[r][k], hp[r][k]) for r in refs for k in range(3)])
        for prec in (1, 2, 3):
            mv = base.copy()
            mv["v"][vec] = mv["v"][vec] << prec
            d_mv = ctx.upload_bytes(mv)
            Pp = dict(P, mv_precision=prec)
            outs = [ctx.plane(w.shape[0], w.shape[1], np.uint8).fill(0x22) for w in want]
            jobs = [sa.obmc_plane(d_mv, Pp, k, hp[refs[0]][k], hp[refs[1] if len(refs) > 1 else refs[0]][k],
                                  res[k], outs[k]) for k in range(3)]
            ctx.obmc_batch(jobs)
            for k in range(3):
                assert np.array_equal(outs[k].download(), want[k]), (n, prec, k)
                outs[k].free()
            d_mv.free()
        for r in refs:
            for p in plain[r] + hp[r]:
                p.free()
        for p in res:
            p.free()


def test_s32_wavelet_on_stream_coefficients(ctx):
    # as the CPU test of the same name: the GPU's s32 inverse wavelet on the stream's real
    # coefficients against the s16 result the digests pin
    import oracle_lib as O
    z = np.load(os.path.join(S.GOLDEN, "stream_pictures.npz"))
    for n in (0, 1, 2):
        depth, wavelet = [int(v) for v in z["p%d_transform" % n]]
        co = [z["p%d_coeff%d" % (n, k)] for k in range(3)]
        src = [ctx.upload(c.astype(np.int32)) for c in co]
        dst = [ctx.plane(c.shape[0], c.shape[1], np.int32) for c in co]
        ctx.iiwt_batch(list(zip(src, dst)), depth, wavelet)
        for k in range(3):
            assert np.array_equal(dst[k].download(), O.inverse_iwt(co[k], depth, wavelet).astype(np.int32)), (n, k)
        for p in src + dst:
            p.free()


def test_equivalent_weights_take_the_general_weight_kernel(ctx):
    # picture weights (2, 2, bits 2) predict exactly what the default (1, 1, bits 1) predicts, but
    # send the launch to the per-pixel general-weight kernel: the stream's pictures must not change
    z = np.load(os.path.join(S.GOLDEN, "stream_pictures.npz"))
    out = {int(z["p%d_number" % n][0]): [z["p%d_out%d" % (n, k)] for k in range(3)] for n in range(8)}
    import oracle_lib as O
    for n in (1, 2):
        tag = "p%d_" % n
        refs = [int(v) for v in z[tag + "number"][4:]]
        depth, wavelet = [int(v) for v in z[tag + "transform"]]
        P = dict(zip(S.PARAM_KEYS, [int(v) for v in z[tag + "params"]]))
        P.update(picture_weight_1=2, picture_weight_2=2, picture_weight_bits=2)
        d_mv = ctx.upload_bytes(z[tag + "mv"])
        keep, jobs, outs = [], [], []
        for k in range(3):
            res = ctx.upload(O.inverse_iwt(z[tag + "coeff%d" % k], depth, wavelet))
            r = [ctx.upload(out[x][k]) for x in refs]
            o = ctx.plane(out[refs[0]][k].shape[0], out[refs[0]][k].shape[1], np.uint8).fill(0x33)
            jobs.append(sa.obmc_plane(d_mv, P, k, r[0], r[1] if len(r) > 1 else r[0], res, o))
            outs.append(o)
            keep += [res, o] + r
        ctx.obmc_batch(jobs)
        for k in range(3):
            assert np.array_equal(outs[k].download(), z[tag + "out%d" % k]), (n, k)
        for p in keep + [d_mv]:
            p.free()
