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
