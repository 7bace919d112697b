#!/usr/bin/env python3
"""Regenerates the low-delay fixtures.

quant_tables.json     -- schro_table_quant[61] and schro_table_offset_1_2[61]: the NUMBERS
    held by /root/reference/schroedinger/schrotables.c (read as text, this container
    only).  Data, no source: two lists of 61 integers.
lowdelay_oracle.npz   -- small pictures: slice bytes + parameters + the coefficient planes
    the C oracle decodes from them (s16 fast / s16 slow / s32 decoders).
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import synth  # noqa: E402

REF_TABLES = "/root/reference/schroedinger/schrotables.c"

# (name, width, height, chroma shifts, depth, slice w, slice h, bytes num, denom, bpp)
CASES = [
    ("fast16", 64, 32, (1, 1), 2, 16, 8, 121, 2, 2),
    ("slow16", 72, 40, (1, 0), 2, 24, 10, 301, 3, 2),
    ("s32", 64, 48, (1, 0), 3, 16, 16, 190, 1, 4),
]


def case_inputs(name, w, h, chroma, depth, sw, sh, num, den, bpp):
    P = synth.lowdelay_params(w, h, chroma, depth, sw, sh, num, den)
    if name == "slow16":        # slice counts that do not divide the LL band
        P["n_horiz_slices"], P["n_vert_slices"] = 5, 3
    q = synth.quantised_planes(P, seed=len(name), scale=1.2, big_every=97,
                               big_range=1 << (15 if bpp == 2 else 22))
    bi = synth.lowdelay_base_index(P, seed=len(name), lo=0, hi=70)
    return P, O.lowdelay_write(q, P, bpp, bi)


def main():
    if os.path.exists(REF_TABLES):
        src = open(REF_TABLES).read()

        def table(name):
            body = re.search(name + r"\[61\]\s*=\s*\{(.*?)\};", src, re.S).group(1)
            return [int(v.rstrip("u")) for v in re.findall(r"\d+u?", body)]
        with open(os.path.join(HERE, "quant_tables.json"), "w") as f:
            json.dump({"schro_table_quant": table("schro_table_quant"),
                       "schro_table_offset_1_2": table("schro_table_offset_1_2")}, f)
    out = {}
    for case in CASES:
        name, bpp = case[0], case[-1]
        P, data = case_inputs(*case)
        planes = [np.zeros((P["iwt_chroma_height"] if k else P["iwt_luma_height"],
                            P["iwt_chroma_width"] if k else P["iwt_luma_width"]),
                           np.int16 if bpp == 2 else np.int32) for k in range(3)]
        O.lowdelay_decode(data, planes, P)
        out[name + "_params"] = np.array([P[k] for k in (
            "transform_depth", "iwt_luma_width", "iwt_luma_height", "iwt_chroma_width", "iwt_chroma_height",
            "n_horiz_slices", "n_vert_slices", "slice_bytes_num", "slice_bytes_denom")] + P["quant_matrix"], np.int32)
        out[name + "_slices"] = data
        for k in range(3):
            out["%s_comp%d" % (name, k)] = planes[k]
    np.savez_compressed(os.path.join(HERE, "lowdelay_oracle.npz"), **out)


if __name__ == "__main__":
    main()
