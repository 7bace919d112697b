"""CPU: the host code and the oracle under the sanitizers (SURVEY 5 "race detection / sanitizers").

* `make -C schroedinger_amd/csrc dry_tsan dry_asan`: every source of the library with -DSCHRO_HIP_DRY
  (schro_hip_dry.h: the HIP runtime's entry points are host stand-ins, kernel launches are dropped) and
  -fsanitize=thread / address,undefined.  The libraries load and run without a device; the sanitizer runtimes are
  preloaded into child Python processes that run tests/test_scheduler.py and tests/dry_run_cases.py (the fuzz file's random geometries through every plane- and frame-layer entry point).
* `make -C oracle asan`: the oracle with gcc's ASAN + UBSAN; its own tests run against it.

A report of any sanitizer fails the test with the report's text.  Never a GPU-side sanitizer run (not available on this
pool)."""
import glob
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "schroedinger_amd", "csrc")
REPORT = re.compile(r"(ThreadSanitizer|AddressSanitizer|LeakSanitizer|UndefinedBehaviorSanitizer|runtime error:)")


def clang_runtime(name):
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.%s-x86_64.so" % name))
    return hits[-1] if hits else None


def run_child(args, env, timeout=900):
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT,
                       env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    text = r.stdout.decode(errors="replace")
    found = REPORT.search(text)
    assert not found, "sanitizer report:\n" + text[max(0, found.start() - 200):found.start() + 4000]
    assert r.returncode == 0, text[-4000:]
    m = re.search(r"(\d+) passed", text)
    return int(m.group(1)) if m else 0


@pytest.mark.timeout(1500)
def test_host_code_under_thread_sanitizer():
    rt = clang_runtime("tsan")
    if not rt:
        pytest.skip("no ThreadSanitizer runtime in this image")
    subprocess.run(["make", "-C", CSRC, "-j8", "-s", "dry_tsan"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = {"SCHRO_HIP_LIB": os.path.join(ROOT, "schroedinger_amd", "libschro_hip_dry_tsan.so"), "LD_PRELOAD": rt,
           "TSAN_OPTIONS": "report_signal_unsafe=0:exitcode=66:halt_on_error=0"}
    # (tests/test_sharding.py stays out: its two gloo ranks run torch's own threads, which this runtime then reports on)
    n = run_child(["tests/test_scheduler.py", "tests/dry_run_cases.py", "-m", "not gpu"], env)
    assert n >= 18, n


@pytest.mark.timeout(1500)
def test_host_code_under_address_and_undefined_behaviour_sanitizers():
    rt = clang_runtime("asan")
    if not rt:
        pytest.skip("no AddressSanitizer runtime in this image")
    subprocess.run(["make", "-C", CSRC, "-j8", "-s", "dry_asan"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = {"SCHRO_HIP_LIB": os.path.join(ROOT, "schroedinger_amd", "libschro_hip_dry_asan.so"), "LD_PRELOAD": rt,
           # (leaks: CPython's own allocations at exit are not ours to report)
           "ASAN_OPTIONS": "detect_leaks=0:exitcode=67", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=0"}
    n = run_child(["tests/test_scheduler.py", "tests/dry_run_cases.py", "-m", "not gpu"], env)
    assert n >= 18, n


@pytest.mark.timeout(1500)
def test_oracle_under_address_and_undefined_behaviour_sanitizers():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE).stdout.decode().strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no gcc AddressSanitizer runtime in this image")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], check=True)
    env = {"SCHRO_ORACLE_LIB": os.path.join(ROOT, "oracle", "libschro_oracle_asan.so"), "LD_PRELOAD": asan,
           "ASAN_OPTIONS": "detect_leaks=0:exitcode=67", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=0"}
    n = run_child(sorted(glob.glob(os.path.join(ROOT, "tests", "test_oracle_*.py"))), env)
    assert n >= 100, n
