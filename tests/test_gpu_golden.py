"""GPU: the committed golden vectors (tests/golden/*.npz).  iiwt_ref_kernels.npz was
produced by the reference's own compiled kernels (see tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

import oracle_lib as O
import schroedinger_amd as sa
import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def run_iiwt(ctx, x, depth, filt):
    src, dst = ctx.upload(x), ctx.plane(x.shape[0], x.shape[1], x.dtype)
    ctx.iiwt_batch([(src, dst)], depth, filt)
    out = dst.download()
    src.free()
    dst.free()
    return out


@pytest.mark.parametrize("name", ["iiwt_ref_kernels.npz", "iiwt_oracle.npz"])
def test_iiwt_golden(ctx, name):
    g = np.load(os.path.join(GOLD, name))
    keys = [k[:-3] for k in g.files if k.endswith("_in")]
    assert keys
    for k in keys:
        filt = int(k.split("_")[1][1:])
        depth = 3 if "_d3" in k else 1
        assert np.array_equal(run_iiwt(ctx, g[k + "_in"], depth, filt), g[k + "_out"]), k


def test_obmc_golden(ctx):
    g = np.load(os.path.join(GOLD, "obmc_oracle.npz"))
    keys = [k[:-4] for k in g.files if k.endswith("_out")]
    assert len(keys) >= 90
    for key in keys:
        parts = key.split("_")
        blk = (int(parts[0][1:]), int(parts[1]))
        prec = int(parts[2][1:])
        weights = (int(parts[3][1:]), int(parts[4]), int(parts[5]))
        chroma = (int(parts[6][1]), int(parts[6][2]))
        k = int(parts[7][1:])
        P = synth.motion_params(64, 48, blk[0], blk[1], prec, weights, chroma)
        mv = g[key + "_mv"].view(sa.MV_DTYPE)
        r1, r2, res, want = g[key + "_r1"], g[key + "_r2"], g[key + "_res"], g[key + "_out"]
        ch, cw = want.shape
        d_mv = ctx.upload_bytes(mv)
        if prec == 0:
            g1, g2 = ctx.upload(r1), ctx.upload(r2)
            tmp = []
        else:
            p1, p2 = ctx.upload(r1), ctx.upload(r2)
            g1, g2 = ctx.hp_plane(ch, cw), ctx.hp_plane(ch, cw)
            ctx.upsample_batch([(p1, g1), (p2, g2)])
            tmp = [p1, p2]
        d_res, out = ctx.upload(res), ctx.plane(ch, cw, np.uint8)
        ctx.obmc_batch([sa.obmc_plane(d_mv, P, k, g1, g2, d_res, out)])
        assert np.array_equal(out.download(), want), key
        for p in tmp + [g1, g2, d_res, out, d_mv]:
            p.free()
