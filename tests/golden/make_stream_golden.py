#!/usr/bin/env python3
"""Decodes the reference's own test stream (testsuite/test_stream.drc, BASELINE config 2) with
the oracle -- oracle/dirac_stream.py for the bitstream, the C oracle for the pixel path -- and
checks the decoded pictures against what the REFERENCE decoder produced for the same stream:
schro_frame_md5 of its first output frames as recorded in SURVEY.md 8(c).  Then writes

arith_lut.json        the 256 numbers of the arithmetic coder's probability LUT and
                      schro_table_offset_3_8[61] as the reference holds them (data, read from
                      /root/reference as text; this container only)
test_stream.drc       the stream itself: a data file of the reference's own testsuite
stream_pictures.npz   for the first pictures (coded order): coefficient planes, motion vector
                      records, parameters, the two references' numbers and the decoded u8
                      picture -- real-stream inputs and expected outputs for the GPU tests
stream_md5.json       schro_frame_md5 of every decoded picture by picture number (the first
                      three are the reference's, the rest are this oracle's)
"""
import json
import os
import re
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_lib as O  # noqa: E402,F401
from stream_lib import decode_stream  # noqa: E402

REF = "/root/reference"
# state words of schro_frame_md5 for the reference decoder's first three output frames
REFERENCE_MD5 = ["5a8b07919a22a6322b7e108c10cc282d", "84e7d9bc8415ddf5a903e02bb0b6cd20",
                 "54d3df001937ac569ed8419da8edaf76"]
N_FIXTURE = 8          # pictures kept as GPU-test fixtures


def reference_tables():
    src = re.sub(r"//.*", "", open(os.path.join(REF, "schroedinger/schroarith.c")).read())
    lut = [int(v) for v in re.findall(r"\d+", re.search(r"static const uint16_t lut\[256\]\s*=\s*\{(.*?)\};", src, re.S).group(1))]
    tsrc = open(os.path.join(REF, "schroedinger/schrotables.c")).read()
    body = re.search(r"schro_table_offset_3_8\[61\]\s*=\s*\{(.*?)\};", tsrc, re.S).group(1)
    off38 = [int(v.rstrip("u")) for v in re.findall(r"\d+u?", body)]
    assert len(lut) == 256 and len(off38) == 61
    return {"arith_lut": lut, "schro_table_offset_3_8": off38}


def main():
    tables = reference_tables()
    with open(os.path.join(HERE, "arith_lut.json"), "w") as f:
        json.dump(tables, f)
    shutil.copyfile(os.path.join(REF, "testsuite/test_stream.drc"), os.path.join(HERE, "test_stream.drc"))
    data = open(os.path.join(HERE, "test_stream.drc"), "rb").read()
    md5, fix = {}, {}
    for n, rec in enumerate(decode_stream(data, tables)):
        md5[rec["number"]] = rec["md5"]
        print("coded %3d picture %3d refs %-10s %s" % (n, rec["number"], rec["refs"], rec["md5"]), flush=True)
        if n < N_FIXTURE:
            tag = "p%d_" % n
            fix[tag + "number"] = np.array([rec["number"], rec["num_refs"], int(rec["is_ref"]), int(rec["zero_residual"])]
                                           + rec["refs"], np.int32)
            for k in range(3):
                fix[tag + "out%d" % k] = rec["out"][k]
                if rec["coeffs"] is not None:
                    fix[tag + "coeff%d" % k] = rec["coeffs"][k]
            if rec["coeffs"] is not None:
                fix[tag + "transform"] = np.array([rec["depth"], rec["wavelet"]], np.int32)
            if rec["num_refs"]:
                fix[tag + "mv"] = rec["mv"]
                fix[tag + "params"] = np.array([rec["params"][k] for k in sorted(rec["params"])], np.int32)
    for k, want in enumerate(REFERENCE_MD5):
        assert md5[k] == want, "picture %d: %s, the reference decoder produced %s" % (k, md5[k], want)
    print("first %d output frames identical to the reference decoder's" % len(REFERENCE_MD5))
    with open(os.path.join(HERE, "stream_md5.json"), "w") as f:
        json.dump({"reference": REFERENCE_MD5, "oracle": [md5[k] for k in sorted(md5)]}, f)
    np.savez_compressed(os.path.join(HERE, "stream_pictures.npz"), **fix)


if __name__ == "__main__":
    main()
