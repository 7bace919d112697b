#!/usr/bin/env python3
"""BASELINE config 2 on one MI355X: the pixel path of the reference's testsuite/test_stream.drc
(100 pictures 320x240 4:2:2; intra DD(9,7), inter LeGall(5,3) depth 4, 12x12/8x8 full-pel OBMC).
The bitstream is decoded once on the CPU by the oracle's front end (entropy decoding is not part
of the path); all coefficients and motion vectors are uploaded; then the 100 pictures run through
inverse wavelet + OBMC / intra convert in coded order (each inter picture predicts from pictures
the GPU decoded before) and the loop is timed.  Every picture is checked against the oracle, whose
first frames carry the reference decoder's digests.  Prints one JSON line.  Lives under tests/
because it needs the oracle: `python3 tests/bench_stream.py`."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import schroedinger_amd as sa  # noqa: E402
import stream_lib as S  # noqa: E402

REPEAT = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def main():
    ctx = sa.Context(0)
    t0 = time.perf_counter()
    recs = list(S.decode_stream(S.load_stream(), S.load_tables()))
    cpu_total = time.perf_counter() - t0        # bitstream + oracle pixel path, one thread
    pics = []
    for rec in recs:
        p = dict(rec=rec)
        p["co"] = [ctx.upload(c) for c in rec["coeffs"]]
        p["res"] = [ctx.plane(c.height, c.width, np.int16) for c in p["co"]]
        p["out"] = [ctx.plane(w.shape[0], w.shape[1], np.uint8) for w in rec["out"]]
        if rec["num_refs"]:
            p["mv"] = ctx.upload_bytes(rec["mv"])
        pics.append(p)
    by_number = {p["rec"]["number"]: p for p in pics}

    def run():
        for p in pics:
            rec = p["rec"]
            ctx.iiwt_batch(list(zip(p["co"], p["res"])), rec["depth"], rec["wavelet"])
            if rec["num_refs"] == 0:
                ctx.convert_u8_batch(list(zip(p["res"], p["out"])))
            else:
                r = [by_number[n]["out"] for n in rec["refs"]]
                ctx.obmc_batch([sa.obmc_plane(p["mv"], rec["params"], k, r[0][k], r[-1][k], p["res"][k], p["out"][k])
                                for k in range(3)])
    run()
    ctx.synchronize()
    ok = all(np.array_equal(p["out"][k].download(), p["rec"]["out"][k]) for p in pics for k in range(3))
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(REPEAT):
        run()
    gpu_ms = ctx.timer_end() / REPEAT
    wall_ms = (time.perf_counter() - t0) * 1e3 / REPEAT
    n = len(pics)
    print(json.dumps({
        "workload": "testsuite/test_stream.drc pixel path, %d pictures 320x240 4:2:2, coded order, one stream" % n,
        "parity_vs_oracle": "bit-exact" if ok else "MISMATCH",
        "gpu_us_per_picture": gpu_ms * 1e3 / n, "host_wall_us_per_picture": wall_ms * 1e3 / n,
        "pictures_per_s": n / (max(gpu_ms, wall_ms) * 1e-3), "Mpix_per_s": n * 320 * 240 / (max(gpu_ms, wall_ms) * 1e3),
        "launches_per_picture": "5 (4 wavelet levels + OBMC) / 5 (intra: 4 + convert)",
        "cpu_oracle_ms_per_picture_incl_bitstream": cpu_total * 1e3 / n}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
