"""CPU: the oracle against the REFERENCE DECODER'S OWN OUTPUT on the reference's own stream.

tests/golden/test_stream.drc is testsuite/test_stream.drc of the reference (BASELINE config 2:
100 pictures 320x240 4:2:2, intra pictures with the DD(9,7) wavelet, inter pictures with
LeGall(5,3), depth 4, 12x12/8x8 OBMC, full-pel vectors, P and B pictures).  SURVEY.md 8(c)
records schro_frame_md5 of the first three frames the reference decoder produced for it.
Decoding the stream with the oracle -- oracle/dirac_stream.py for the bitstream, the C oracle
for inverse wavelet, OBMC from one and two references, residual add and clamp -- gives the
same three digests: for this path the oracle is pinned by reference output, not only by
reference kernels."""
import json
import os

import numpy as np

import stream_lib as S


def test_first_frames_match_the_reference_decoder():
    md5 = json.load(open(os.path.join(S.GOLDEN, "stream_md5.json")))
    assert md5["reference"] == ["5a8b07919a22a6322b7e108c10cc282d", "84e7d9bc8415ddf5a903e02bb0b6cd20",
                                "54d3df001937ac569ed8419da8edaf76"]
    got = {}
    # coded order I0 P3 B1 B2: picture 3 is predicted from 0, pictures 1 and 2 from 0 and 3
    for rec in S.decode_stream(S.load_stream(), S.load_tables(), limit=4):
        got[rec["number"]] = rec["md5"]
    assert [got[k] for k in (0, 1, 2)] == md5["reference"]
    assert got[3] == md5["oracle"][3]


def test_fixture_pictures_are_what_the_oracle_decodes():
    z = np.load(os.path.join(S.GOLDEN, "stream_pictures.npz"))
    md5 = json.load(open(os.path.join(S.GOLDEN, "stream_md5.json")))["oracle"]
    for n, rec in enumerate(S.decode_stream(S.load_stream(), S.load_tables(), limit=8)):
        tag = "p%d_" % n
        assert z[tag + "number"][0] == rec["number"]
        assert rec["md5"] == md5[rec["number"]]
        for k in range(3):
            assert np.array_equal(z[tag + "out%d" % k], rec["out"][k])
            assert np.array_equal(z[tag + "coeff%d" % k], rec["coeffs"][k])
        if rec["num_refs"]:
            assert np.array_equal(z[tag + "mv"], rec["mv"])


def test_scaled_vectors_through_the_sub_pel_path():
    # The stream is full-pel, so its reference-pinned pictures say nothing about the half-pel
    # images or the sub-pel fetch.  But a picture rendered with mv_precision p and its vectors
    # multiplied by 2^p must come out identical (integer positions of the upsampled reference
    # are the reference's own samples, the blend degenerates to a copy): that runs the real
    # stream's blocks through get_block's prec 1 / 2 / 3 branches and the upsampled planes.
    import oracle_lib as O
    z = np.load(os.path.join(S.GOLDEN, "stream_pictures.npz"))
    out = {int(z["p%d_number" % n][0]): [z["p%d_out%d" % (n, k)] for k in range(3)] for n in range(8)}
    for n in (1, 2, 5):                                  # a P picture and two B pictures
        tag = "p%d_" % n
        refs = [int(v) for v in z[tag + "number"][4:]]
        depth, wavelet = [int(v) for v in z[tag + "transform"]]
        P = dict(zip(S.PARAM_KEYS, [int(v) for v in z[tag + "params"]]))
        res = [O.inverse_iwt(z[tag + "coeff%d" % k], depth, wavelet) for k in range(3)]
        # chroma vectors are the luma vectors >> the chroma shift (4:2:2 here): keep dx even so
        # that scaling the vector and shifting it commute, and take the full-pel rendering of
        # these vectors (the reference-pinned code path) as the expectation
        base = z[tag + "mv"].copy()
        vec = (base["flags"] & 3) != 0                   # DC blocks keep their DC values
        base["v"][vec, 0:2] &= ~1
        want = []
        for k in range(3):
            u = [O.UpComp(out[r][k], upsample=False) for r in refs]
            shape = z[tag + "out%d" % k].shape
            want.append(O.motion_render(base, O.MotionParams(**P), k, u[0], u[1] if len(u) > 1 else None,
                                        res[k], shape[1], shape[0]))
        for prec in (1, 2, 3):
            mv = base.copy()
            mv["v"][vec] = mv["v"][vec] << prec
            op = O.MotionParams(**dict(P, mv_precision=prec))
            for k in range(3):
                u = [O.UpComp(out[r][k], upsample=True) for r in refs]
                got = O.motion_render(mv, op, k, u[0], u[1] if len(u) > 1 else None, res[k],
                                      want[k].shape[1], want[k].shape[0])
                assert np.array_equal(got, want[k]), (n, prec, k)


def test_s32_wavelet_on_stream_coefficients():
    # the s32 kernels (schroorc.orc:1815-2164) on real coefficients, whose values stay far inside
    # 16 bits at every lifting step of these two filters, must agree with the s16 kernels that
    # the stream digests pin
    import oracle_lib as O
    z = np.load(os.path.join(S.GOLDEN, "stream_pictures.npz"))
    for n in (0, 1, 2):
        depth, wavelet = [int(v) for v in z["p%d_transform" % n]]
        for k in range(3):
            co = z["p%d_coeff%d" % (n, k)]
            want = O.inverse_iwt(co, depth, wavelet)
            got = O.inverse_iwt(co.astype(np.int32), depth, wavelet)
            assert np.array_equal(got, want.astype(np.int32)), (n, k)


def test_equivalent_weights_take_the_general_weight_arithmetic():
    # picture_weight (2, 2, bits 2) is the same prediction as the default (1, 1, bits 1), but the
    # reference computes it with its general-weight block arithmetic (schromotion8.c:391-397,
    # 44-73, 621-650) instead of avgub: the stream's pictures must not change
    import oracle_lib as O
    z = np.load(os.path.join(S.GOLDEN, "stream_pictures.npz"))
    out = {int(z["p%d_number" % n][0]): [z["p%d_out%d" % (n, k)] for k in range(3)] for n in range(8)}
    for n in (1, 2):
        tag = "p%d_" % n
        refs = [int(v) for v in z[tag + "number"][4:]]
        depth, wavelet = [int(v) for v in z[tag + "transform"]]
        P = dict(zip(S.PARAM_KEYS, [int(v) for v in z[tag + "params"]]))
        P.update(picture_weight_1=2, picture_weight_2=2, picture_weight_bits=2)
        for k in range(3):
            res = O.inverse_iwt(z[tag + "coeff%d" % k], depth, wavelet)
            u = [O.UpComp(out[r][k], upsample=False) for r in refs]
            want = z[tag + "out%d" % k]
            got = O.motion_render(z[tag + "mv"], O.MotionParams(**P), k, u[0], u[1] if len(u) > 1 else None,
                                  res, want.shape[1], want.shape[0])
            assert np.array_equal(got, want), (n, k)
