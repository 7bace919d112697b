"""bench.py --gpus N starts N rank processes by itself.

CPU-only checks: the launcher spawns exactly N children with the rank environment of the
driver's torchrun form, propagates their exit codes, and on a box without a HIP device every
rank fails loudly instead of one process reporting n_gpus 1.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_spawn_ranks_environment():
    import bench
    calls = []

    class P:
        def __init__(self, cmd, env=None, stdout=None):
            calls.append((cmd, env, stdout))
            self.rc = 3 if env["RANK"] == "1" else 0

        def wait(self):
            return self.rc

    rc = bench.spawn_ranks(4, ["--gpus", "4", "--steps", "2"], popen=P)
    assert rc == 3                                  # worst child code comes back
    assert len(calls) == 4
    ports = set()
    for r, (cmd, env, stdout) in enumerate(calls):
        assert cmd[0] == sys.executable and cmd[1].endswith("bench.py")
        assert cmd[2:] == ["--gpus", "4", "--steps", "2"]
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1"
        ports.add(env["MASTER_PORT"])
        assert (stdout is None) == (r == 0)         # only rank 0 prints the JSON line
    assert len(ports) == 1


def _has_gpu():
    try:
        import schroedinger_amd as sa
        return sa.device_count() > 0
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the behaviour of a box without a HIP device")
def test_gpus_2_fails_loudly_without_devices():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert '"n_gpus"' not in p.stdout               # no bench line from a run that had no GPUs
    assert p.stderr.count("needs a HIP device") >= 2 or p.stderr.count("rank") >= 2, p.stderr
