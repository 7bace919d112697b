#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

iiwt_ref_kernels.npz  -- inputs and outputs of ONE inverse wavelet level computed by
    the REFERENCE's own compiled kernels (oracle/_ref/libschroorc_ref.so, i.e.
    /root/reference/schroedinger/schroorc-dist.c built unmodified) driven in the
    reference's row schedule by oracle/ref_driver.c.  Filters 0-4 and 6, s16 and s32,
    legal-range and full-range inputs.  Needs /root/reference (this container only).
iiwt_oracle.npz       -- the same cases plus filter 5 and multi-level cases from the
    C oracle (oracle/libschro_oracle.so); filter 5 has no Orc kernel in the reference.
obmc_oracle.npz       -- small OBMC cases (inputs + expected u8 output) from the C oracle.

Fixtures are data only (inputs + expected outputs); no reference source is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import synth  # noqa: E402

SIZES = [(2, 2), (4, 6), (16, 16), (18, 34), (64, 64)]


def main():
    ref, ora = {}, {}
    for dt in (np.int16, np.int32):
        tag = "s16" if dt == np.int16 else "s32"
        for f in range(7):
            for (h, w) in SIZES:
                img = synth.image_s(h, w, dt, seed=h * 31 + w + f)
                legal = O.iwt_2d(img, f)
                full = synth.full_range(h, w, dt, seed=h * 17 + w + f)
                for kind, x in (("legal", legal), ("full", full)):
                    key = "%s_f%d_%dx%d_%s" % (tag, f, h, w, kind)
                    ora[key + "_in"] = x
                    ora[key + "_out"] = O.iiwt_2d(x, f)
                    if f != 5 and O.ref_available():
                        ref[key + "_in"] = x
                        ref[key + "_out"] = O.refdrv_iiwt_2d(x, f)
            img = synth.image_s(48, 64, dt, seed=77 + f)
            co = O.forward_iwt(img, 3, f)
            ora["%s_f%d_48x64_d3_in" % (tag, f)] = co
            ora["%s_f%d_48x64_d3_out" % (tag, f)] = O.inverse_iwt(co, 3, f)
    if ref:
        np.savez_compressed(os.path.join(HERE, "iiwt_ref_kernels.npz"), **ref)
    np.savez_compressed(os.path.join(HERE, "iiwt_oracle.npz"), **ora)

    ob = {}
    n = 0
    for blk in ((8, 4), (12, 8), (16, 12), (24, 16)):
        for prec in range(4):
            for weights in ((1, 1, 1), (2, 3, 1), (3, 5, 3)):
                chroma = [(1, 1), (1, 0), (0, 0)][n % 3]
                n += 1
                w, h = 64, 48
                P = synth.motion_params(w, h, blk[0], blk[1], prec, weights, chroma)
                mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 40 << prec, seed=n)
                for k in (0, 1):
                    cw = w if k == 0 else -(-w // (1 << chroma[0]))
                    ch = h if k == 0 else -(-h // (1 << chroma[1]))
                    r1 = synth.picture_u8(ch, cw, seed=100 + n)
                    r2 = synth.picture_u8(ch, cw, seed=200 + n)
                    res = synth.image_s(ch, cw, np.int16, seed=300 + n)
                    out = O.motion_render(mv, O.MotionParams(**P), k, O.UpComp(r1, upsample=prec > 0),
                                          O.UpComp(r2, upsample=prec > 0), res, cw, ch)
                    key = "b%d_%d_p%d_w%d_%d_%d_c%d%d_k%d" % (blk + (prec,) + weights + chroma + (k,))
                    ob[key + "_mv"] = mv.view(np.uint8)
                    ob[key + "_r1"], ob[key + "_r2"], ob[key + "_res"], ob[key + "_out"] = r1, r2, res, out
    np.savez_compressed(os.path.join(HERE, "obmc_oracle.npz"), **ob)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
