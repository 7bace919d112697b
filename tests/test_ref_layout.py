"""The frame-layer structs of include/schro_hip.h are layout-identical to the reference's
SchroFrameData / SchroFrame / SchroParams / SchroMemoryDomain (the boundary must
match the interface it cites).  tests/golden/ref_layout.json holds the reference's offsets,
recorded from its own headers by scripts/ref_layout.py; here a C program built with OUR header
prints the mirror structs' offsets and the two are compared member by member.  Where the
reference's headers are present (the build container), tests/c/layout_check.c also compiles:
a _Static_assert per member against the real structs."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIRROR = {"SchroFrameData": "SchroHipFrameData", "SchroFrame": "SchroHipFrame",
          "SchroParams": "SchroHipParams", "SchroMemoryDomain": "SchroHipMemoryDomain",
          "SchroMotion": "SchroHipMotion"}


def test_mirror_structs_have_the_reference_layout(tmp_path):
    lay = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_layout.json")))
    src = ['#include "schro_hip.h"', "#include <stddef.h>", "#include <stdio.h>", "int main (void) {"]
    for t, members in lay.items():
        for m in members:
            if m == "sizeof":
                if t != "SchroMemoryDomain":        # ours has private members after `slots`
                    src.append('printf ("%s sizeof %%zu\\n", sizeof (%s));' % (t, MIRROR[t]))
                else:
                    src.append('printf ("%s sizeof %%zu\\n", offsetof (%s, ctx));' % (t, MIRROR[t]))
            elif not m.startswith("_"):
                src.append('printf ("%s %s %%zu\\n", offsetof (%s, %s));' % (t, m, MIRROR[t], m))
    src.append("return 0; }")
    c, exe = tmp_path / "mirror.c", tmp_path / "mirror"
    c.write_text("\n".join(src))
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    got = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        t, m, v = line.split()
        got.setdefault(t, {})[m] = int(v)
    for t, members in lay.items():
        for m, off in members.items():
            if not m.startswith("_"):
                assert got[t][m] == off, "%s.%s: header %d, reference %d" % (t, m, got[t][m], off)


@pytest.mark.skipif(not os.path.isdir("/root/reference/schroedinger"), reason="reference headers not here")
def test_static_asserts_against_the_reference_headers(tmp_path):
    subprocess.check_call(["gcc", "-I/root/reference", "-I" + os.path.join(ROOT, "include"), "-c",
                           os.path.join(ROOT, "tests", "c", "layout_check.c"), "-o", str(tmp_path / "lc.o")])
