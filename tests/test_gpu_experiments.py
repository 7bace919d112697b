"""GPU: the EXPERIMENTS build of the library (schroedinger_amd/libschro_hip_exp.so: the same sources with
-DSCHRO_HIP_EXPERIMENTS) in child processes.

The product library reads no SCHRO_HIP_* switch and does not contain the measured-slower formulations (the LDS-fused
wavelet group, the chain form of the register wavelet); the experiments build has both, and with its switches every
kernel the library chooses between can be FORCED where another would be chosen -- so each formulation is compared
with the same oracle on the whole case matrix of its test file:

  * no switch: the tests of test_gpu_iiwt.py / test_gpu_lowdelay.py that set switches themselves (fused levels,
    small / large register tiles, LDS kernel instead of the register kernel, per-level Haar, slice_kernel instead of
    slice_run_kernel, dc_predict_kernel instead of dc_skew_kernel) -- in the product library those switches are inert;
  * SCHRO_HIP_OBMC_KERNEL=item: obmc.hip's item kernel for every default-weight case (also out of pair images);
  * SCHRO_HIP_OBMC_MERGE=2: U + V planes of one-component images as one job (obmc_row_kernel_*_2) always;
  * SCHRO_HIP_IIWT_CHAIN=1: every level of the register wavelet in one launch (r04, iiwt_reg.hip);
  * SCHRO_HIP_OBMC_STRIP=1: the 12 / 8 block set's luma planes by the strip kernel (r05, obmc_strip.hip: accumulator in
    registers, no LDS tile -- a third formulation of the same arithmetic; measured 2.5 x slower);
  * SCHRO_HIP_UPSAMPLE_PERSIST=n: the upsample as n persistent workgroups per CU that prefetch the next tile (r06);
  * SCHRO_HIP_OBMC_PAD=1: the prediction-only 12-pixel-row OBMC kernel with line-aligned quads of lanes (r06).
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "schroedinger_amd", "libschro_hip_exp.so")


def run(files, env=None, k=None, timeout=900):
    assert os.path.exists(EXP), "build the experiments library first (__graft_entry__.build ())"
    e = dict(os.environ, SCHRO_HIP_LIB=EXP, **(env or {}))
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu"] + [os.path.join(ROOT, "tests", f) for f in files]
    if k:
        cmd += ["-k", k]
    p = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    return p.stdout


def test_switch_driven_tests_with_live_switches():
    run(["test_gpu_iiwt.py", "test_gpu_lowdelay.py"])


def test_item_kernel_takes_every_default_weight_case():
    run(["test_gpu_obmc.py"], env={"SCHRO_HIP_OBMC_KERNEL": "item"},
        k="test_default_weights or test_dc_values or test_rotating or test_pair_images_default or test_pair_images_edges")


def test_u_and_v_planes_always_one_job():
    run(["test_gpu_obmc.py"], env={"SCHRO_HIP_OBMC_MERGE": "2"},
        k="test_default_weights or test_dc_values or test_rotating or test_ragged")


def test_register_wavelet_chain_form():
    out = run(["test_gpu_iiwt.py", "test_gpu_stream.py", "test_gpu_fuzz.py"], env={"SCHRO_HIP_IIWT_CHAIN": "1"})
    assert "passed" in out


def test_strip_kernel_takes_the_12_8_luma_planes():
    out = run(["test_gpu_obmc.py", "test_gpu_combine.py", "test_gpu_stream.py", "test_gpu_fuzz.py"], env={"SCHRO_HIP_OBMC_STRIP": "1"})
    assert "passed" in out


def test_persistent_prefetching_upsample():
    """r06 (measured, not the product's form): a grid of one workgroup per CU that loops over the
    tiles and asks for the next tile's source before it filters and stores the current one -- the same planes."""
    out = run(["test_gpu_frameops.py", "test_gpu_stream.py"], env={"SCHRO_HIP_UPSAMPLE_PERSIST": "1"})
    assert "passed" in out


def test_line_aligned_quads_in_the_obmc_passes():
    """r06 (measured slower, HISTORY 9): the prediction-only 12-pixel-row kernel with every block's lanes laid out as whole
    128-byte lines of its reference (masked lanes in front of and behind the window's rows) -- the same pictures."""
    out = run(["test_gpu_combine.py", "test_gpu_stages.py", "test_gpu_fuzz.py"], env={"SCHRO_HIP_OBMC_PAD": "1"})
    assert "passed" in out

