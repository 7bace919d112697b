#!/usr/bin/env python3
"""Record the layout of the reference's boundary structs from ITS OWN headers.

Run in the build container (needs /root/reference, gcc): compiles a small offsetof dumper
against /root/reference/schroedinger/{schroframe,schroparams,schrodomain}.h -- which compile as
they are (no liborc, no stand-in) -- and writes tests/golden/ref_layout.json: plain numbers, the
fixture that pins the mirror structs of include/schro_hip.h (tests/test_ref_layout.py).
SchroMotion (schromotion.h) pulls in <orc/orc.h> and cannot be compiled here; its first four
members are pointers (offsets 0, 8, 16, 24 by the ABI) and are recorded as such."""
import json
import os
import subprocess
import sys
import tempfile

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MEMBERS = {
    "SchroFrameData": ["format", "data", "stride", "width", "height", "length", "h_shift", "v_shift"],
    "SchroFrame": ["refcount", "free", "domain", "regions", "priv", "format", "width", "height", "components",
                   "is_virtual", "cached_lines", "virt_frame1", "virt_frame2", "render_line", "virt_priv",
                   "virt_priv2", "extension", "cache_offset", "is_upsampled", "upsample_done"],
    "SchroParams": ["video_format", "is_noarith", "wavelet_filter_index", "transform_depth", "horiz_codeblocks",
                    "vert_codeblocks", "codeblock_mode_index", "num_refs", "have_global_motion", "xblen_luma",
                    "yblen_luma", "xbsep_luma", "ybsep_luma", "mv_precision", "global_motion", "picture_pred_mode",
                    "picture_weight_bits", "picture_weight_1", "picture_weight_2", "is_lowdelay", "n_horiz_slices",
                    "n_vert_slices", "slice_bytes_num", "slice_bytes_denom", "quant_matrix", "iwt_chroma_width",
                    "iwt_chroma_height", "iwt_luma_width", "iwt_luma_height", "x_num_blocks", "y_num_blocks",
                    "x_offset", "y_offset"],
    "SchroMemoryDomain": ["mutex", "flags", "alloc", "alloc_2d", "free", "slots"],
}


def main():
    src = ["#define SCHRO_ENABLE_UNSTABLE_API", "#include <schroedinger/schroframe.h>",
           "#include <schroedinger/schroparams.h>", "#include <schroedinger/schrodomain.h>",
           "#include <stddef.h>", "#include <stdio.h>", "int main (void) {"]
    for t, ms in MEMBERS.items():
        src.append('printf ("%s sizeof %%zu\\n", sizeof (%s));' % (t, t))
        for m in ms:
            src.append('printf ("%s %s %%zu\\n", offsetof (%s, %s));' % (t, m, t, m))
    src += ["return 0; }"]
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "lay.c"), os.path.join(d, "lay")
        open(c, "w").write("\n".join(src))
        subprocess.check_call(["gcc", "-I" + REF, c, "-o", exe])
        out = subprocess.check_output([exe], text=True)
    lay = {}
    for line in out.splitlines():
        t, m, v = line.split()
        lay.setdefault(t, {})[m] = int(v)
    lay["SchroMotion"] = {"src1": 0, "src2": 8, "motion_vectors": 16, "params": 24,
                          "_note": "pointer members by the LP64 ABI; schromotion.h needs <orc/orc.h>"}
    path = os.path.join(ROOT, "tests", "golden", "ref_layout.json")
    json.dump(lay, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
