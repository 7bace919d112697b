"""Two in-order queues per context (schro_hip_context_select_queue / _queue_mark /
_queue_wait_mark): picture batch k's OBMC on queue 1 beside batch k + 1's inverse wavelet on
queue 0, with the decoder's stage dependencies as marks.  Every picture of every batch must
equal the oracle's, whatever the interleaving on the device."""
import numpy as np
import pytest

import oracle_lib as O
import schroedinger_amd as sa
import synth

pytestmark = pytest.mark.gpu


def test_pipelined_batches_equal_the_oracle(ctx):
    w, h, depth, filt, prec = 320, 256, 3, 0, 2
    P = synth.motion_params(w, h, 12, 8, prec, (1, 1, 1), (1, 1))
    op = O.MotionParams(**P)
    dims = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
    refs = [[synth.picture_u8(hh, ww, seed=5 + 10 * r + k) for k, (hh, ww) in enumerate(dims)] for r in range(2)]
    ups = [[O.UpComp(p, upsample=True) for p in comps] for comps in refs]
    d_ref = [[ctx.upload(p) for p in comps] for comps in refs]
    nsets, steps = 2, 6
    sets = []
    for s in range(nsets):
        hp = [[ctx.hp_plane(hh, ww) for (hh, ww) in dims] for _ in range(2)]
        res = [ctx.plane(hh, ww, np.int16) for (hh, ww) in dims]
        out = [ctx.plane(hh, ww, np.uint8) for (hh, ww) in dims]
        sets.append(dict(hp=hp, res=res, out=out))
    # every step has its own coefficients and vectors; a set's frames are reused every nsets steps
    coeffs = [[synth.image_s(hh, ww, np.int16, seed=100 + 3 * k + c) >> 4 for c, (hh, ww) in enumerate(dims)]
              for k in range(steps)]
    mvs = [synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 48, seed=200 + k) for k in range(steps)]
    d_co = [[ctx.upload(c) for c in cs] for cs in coeffs]
    d_mv = [ctx.upload_bytes(m) for m in mvs]
    got = []
    for k in range(steps):
        s = k % nsets
        b = sets[s]
        ctx.select_queue(0)
        ctx.queue_wait_mark(8 + s)
        ctx.upsample_batch([(d_ref[r][c], b["hp"][r][c]) for r in range(2) for c in range(3)])
        ctx.iiwt_batch([(d_co[k][c], b["res"][c]) for c in range(3)], depth, filt)
        ctx.queue_mark(s)
        ctx.select_queue(1)
        ctx.queue_wait_mark(s)
        ctx.obmc_batch([sa.obmc_plane(d_mv[k], P, c, b["hp"][0][c], b["hp"][1][c], b["res"][c], b["out"][c])
                        for c in range(3)])
        # the pictures leave on the render queue, behind their OBMC (and before the set is reused)
        got.append([b["out"][c].download() for c in range(3)])
        ctx.queue_mark(8 + s)
    ctx.select_queue(0)
    ctx.synchronize()
    for k in range(steps):
        for c, (hh, ww) in enumerate(dims):
            res = O.inverse_iwt(coeffs[k][c], depth, filt)
            want = O.motion_render(mvs[k], op, c, ups[0][c], ups[1][c], res, ww, hh)
            assert np.array_equal(got[k][c], want), "step %d component %d" % (k, c)


def test_queue_arguments_are_checked(ctx):
    with pytest.raises(sa.SchroHipError):
        ctx.select_queue(4)         # queues 0, 1 (kernels), 2, 3 (copies)
    with pytest.raises(sa.SchroHipError):
        ctx.queue_mark(16)
    ctx.queue_wait_mark(7)      # never recorded: no-op
    ctx.select_queue(0)


def test_queues_on_disjoint_compute_units():
    # schro_hip_queue_set_cu_mask: queue 0 on every fourth CU, queue 1 on the others -- the inverse wavelet on
    # one, the same on the other, both equal to the oracle's (a mask changes where a kernel runs, not what it does);
    # a context of its own: the masks stay with its queues
    ctx = sa.Context(0)
    try:
        q0 = [1 if i % 4 == 0 else 0 for i in range(256)]
        ctx.queue_set_cu_mask(0, q0)
        ctx.queue_set_cu_mask(1, [1 - b for b in q0])
        h, w, depth, filt = 144, 208, 3, 0
        coeff = synth.image_s(h, w, np.int16, seed=77) >> 3
        want = O.inverse_iwt(coeff, depth, filt)
        outs = []
        for q in (0, 1, 0):
            ctx.select_queue(q)
            src, dst = ctx.upload(coeff), ctx.plane(h, w, np.int16)
            ctx.iiwt_batch([(src, dst)], depth, filt)
            outs.append(dst)
        ctx.select_queue(0)
        ctx.synchronize()
        for n, dst in enumerate(outs):
            assert np.array_equal(dst.download(), want), n
        with pytest.raises(sa.SchroHipError):
            ctx.queue_set_cu_mask(7, q0)
    finally:
        ctx.close()
