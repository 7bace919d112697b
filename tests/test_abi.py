"""CPU: the C-ABI library exists, loads without a GPU, and exports exactly the entry
points include/schro_hip.h declares; struct layouts seen from Python match the header."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "schro_hip.h")


def header_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(schro_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from schroedinger_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build libschro_hip.so first (__graft_entry__.build())"
    lib = _lib.load()
    declared = header_functions()
    assert len(declared) >= 30
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared


def test_product_library_reads_no_environment_switch_but_debug():
    """r04 hygiene: the switches between kernel forms and the measured-slower forms live in libschro_hip_exp.so
    (-DSCHRO_HIP_EXPERIMENTS) only; the product library's one environment variable is SCHRO_HIP_DEBUG."""
    from schroedinger_amd import _lib
    def env_names(path):
        blob = open(path, "rb").read()
        return sorted(set(m.decode() for m in re.findall(rb"SCHRO_HIP_[A-Z0-9_]{3,}", blob)))
    assert env_names(_lib.LIB_PATH) == ["SCHRO_HIP_DEBUG"]
    exp = os.path.join(os.path.dirname(_lib.LIB_PATH), "libschro_hip_exp.so")
    if os.path.exists(exp):
        names = env_names(exp)
        assert "SCHRO_HIP_IIWT_CHAIN" in names and "SCHRO_HIP_OBMC_KERNEL" in names, names


def test_no_gpu_means_loud_failure_not_fallback():
    import schroedinger_amd as sa
    if sa.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sa.SchroHipError):
        sa.Context(0)


def test_struct_layouts_match_header(tmp_path):
    # compile a tiny C program against the header and compare sizeof/offsetof with ctypes
    from schroedinger_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "schro_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(SchroHipIwtPlane), sizeof(SchroHipObmcPlane),
         sizeof(SchroHipFrameData), sizeof(SchroHipFrame), sizeof(SchroHipParams), sizeof(SchroHipMotion),
         offsetof(SchroHipObmcPlane, residual), offsetof(SchroHipFrame, components));
  return 0;
}''')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = list(map(int, subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()))
    want = [C.sizeof(_lib.IwtPlane), C.sizeof(_lib.ObmcPlane), C.sizeof(_lib.FrameData), C.sizeof(_lib.Frame),
            C.sizeof(_lib.Params), C.sizeof(_lib.Motion), _lib.ObmcPlane.residual.offset,
            _lib.Frame.components.offset]
    assert got == want


def test_motion_vector_record_is_20_bytes():
    import schroedinger_amd as sa
    assert sa.MV_DTYPE.itemsize == 20          # SchroMotionVector, schromotion.h:20-37
    assert sa.MV_DTYPE.fields["v"][1] == 12


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "schroedinger_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                t = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r'#include\s*["<][^">]*oracle|import\s+oracle|from\s+oracle|libschro_oracle|dlopen', t):
                    bad.append(f)
    assert not bad, bad


def test_no_compile_time_knobs_outside_the_experiments_guard():
    """VERDICT r05 hygiene: the product sources carry no scratch switches -- every preprocessor conditional of
    schroedinger_amd/csrc is the experiments guard (SCHRO_HIP_EXPERIMENTS), the sanitizers' device-free build
    (SCHRO_HIP_DRY, schro_hip_dry.h: test infrastructure), the include guard / __HIPCC__ split of the shared header, or
    C++ feature plumbing; an A/B form lives under the experiments guard or is deleted."""
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "schroedinger_amd", "csrc")
    allowed = re.compile(r"^\s*#\s*(ifdef|ifndef|if)\s+(defined\s*\(?\s*)?(SCHRO_HIP_EXPERIMENTS|SCHRO_HIP_DRY|__HIPCC__|__cplusplus)\b")
    bad = []
    for path in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.cpp")) + glob.glob(os.path.join(root, "*.h"))):
        for n, line in enumerate(open(path), 1):
            if re.match(r"^\s*#\s*(ifdef|ifndef|if)\b", line) and not allowed.match(line):
                bad.append("%s:%d: %s" % (os.path.basename(path), n, line.strip()))
    assert not bad, bad
