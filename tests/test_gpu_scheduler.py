"""GPU: the reference's own stream decoded through the multi-device scheduler
(schro_hip_scheduler_*) on the devices of this box (one here): every picture's pixel path runs
as the picture function on the exec-domain thread of the device that owns its reference chain,
and every decoded picture carries the oracle's digest."""
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


def test_stream_through_the_scheduler():
    md5 = json.load(open(os.path.join(S.GOLDEN, "stream_md5.json")))["oracle"]
    sched = sa.Scheduler(0)                 # every visible device
    assert sched.n_devices >= 1
    decoded = [dict() for _ in range(sched.n_devices)]      # per device: picture number -> u8 planes
    results = {}

    def picture(rec):
        def run(ctx, dev):
            want = rec["out"]
            out = [ctx.plane(w.shape[0], w.shape[1], np.uint8) for w in want]
            if rec["zero_residual"]:
                res = [ctx.upload(np.zeros(w.shape, np.int16)) for w in want]
            else:
                co = [ctx.upload(c) for c in rec["coeffs"]]
                res = [ctx.plane(c.height, c.width, np.int16) for c in co]
                ctx.iiwt_batch(list(zip(co, res)), rec["depth"], rec["wavelet"])
            if rec["num_refs"] == 0:
                ctx.convert_u8_batch(list(zip(res, out)))
            else:
                refs = rec["refs"]
                d_mv = ctx.upload_bytes(rec["mv"])
                mine = decoded[dev]             # the references live on THIS device by construction
                ctx.obmc_batch([sa.obmc_plane(d_mv, rec["params"], k, mine[refs[0]][k],
                                              mine[refs[1] if len(refs) > 1 else refs[0]][k], res[k], out[k])
                                for k in range(3)])
            results[rec["number"]] = [o.download() for o in out]
            if rec["is_ref"]:
                decoded[dev][rec["number"]] = out
            return 0
        return run

    wants, n = {}, 0
    for rec in S.decode_stream(S.load_stream(), S.load_tables(), limit=30):
        dev, foreign = sched.submit(rec["number"], rec["refs"], rec["is_ref"], picture(rec))
        assert foreign == -1 and 0 <= dev < sched.n_devices
        wants[rec["number"]] = rec["out"]
        n += 1
    assert sched.wait() == 0
    assert len(results) == n
    for number, got in results.items():
        for k in range(3):
            assert np.array_equal(got[k], wants[number][k]), (number, k)
        assert D.frame_md5(got) == md5[number]
    sched.close()
