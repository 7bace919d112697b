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


def _fake_sysfs(root, cards, nodes):
    """cards: [(card name, pci address, vendor, numa_node)]; nodes: {node: cpulist text}."""
    for name, pci, vendor, node in cards:
        dev = os.path.join(root, "devices", "pci0000:00", pci)
        os.makedirs(dev, exist_ok=True)
        open(os.path.join(dev, "vendor"), "w").write(vendor + "\n")
        open(os.path.join(dev, "numa_node"), "w").write("%d\n" % node)
        os.makedirs(os.path.join(root, "class", "drm"), exist_ok=True)
        os.symlink(dev, os.path.join(root, "class", "drm", name + "_dev"))
        os.makedirs(os.path.join(root, "class", "drm", name))
        os.symlink(dev, os.path.join(root, "class", "drm", name, "device"))
    for node, text in nodes.items():
        d = os.path.join(root, "devices", "system", "node", "node%d" % node)
        os.makedirs(d)
        open(os.path.join(d, "cpulist"), "w").write(text + "\n")


def test_rank_binds_to_its_gpus_numa_node(tmp_path):
    """r05: before the first HIP call a rank of an N > 1 run binds its host threads to the NUMA
    node of its GPU -- from sysfs alone (drm cards of vendor 0x1002 in PCI address order)."""
    import bench
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    root = str(tmp_path)
    _fake_sysfs(root,
                # (card numbers do not follow the PCI order; card0 is some other vendor's display adapter)
                [("card0", "0000:03:00.0", "0x1a03", 0), ("card2", "0000:85:00.0", "0x1002", 1),
                 ("card1", "0000:05:00.0", "0x1002", 0), ("card3", "0000:c5:00.0", "0x1002", -1)],
                {0: "0-7,16-23", 1: "8-15,24-31"})
    assert bench.gpu_numa_cpus(0, root) == (0, list(range(0, 8)) + list(range(16, 24)))
    assert bench.gpu_numa_cpus(1, root)[0] == 1
    assert bench.gpu_numa_cpus(2, root) == (None, [])          # numa_node -1: the box does not say
    assert bench.gpu_numa_cpus(3, root) == (None, [])          # no such GPU
    got = []
    r = bench.bind_to_gpu_numa_node(1, root, setaffinity=got.append, getaffinity=lambda: set(range(0, 12)))
    assert r == {"numa_node": 1, "bound": True, "cpus": 4} and got == [[8, 9, 10, 11]]
    got = []
    r = bench.bind_to_gpu_numa_node(1, root, setaffinity=got.append, getaffinity=lambda: {0, 1})
    assert r["bound"] is False and got == []                    # nothing of that node in the mask: left alone
    assert bench.bind_to_gpu_numa_node(2, root, setaffinity=got.append)["bound"] is False


def test_coherent_motion_field_keeps_the_mode_mix():
    import numpy as np
    import bench
    import synth
    nbx, nby = 480, 272
    a, b = synth.motion_field(nbx, nby, 64, seed=9000), bench.coherent_motion_field(nbx, nby, 9000)
    assert np.array_equal(a["flags"], b["flags"])
    dc = (a["flags"] & 3) == 0
    assert np.array_equal(a["v"][dc], b["v"][dc])
    v = b["v"][~dc].astype(int)
    # neighbours differ by the noise (+-2 quarter-pels) and the zoom's step, not by +-64
    d = np.abs(np.diff(b["v"].astype(int).reshape(nby, nbx, 4)[:, :, 0], axis=1))[~dc.reshape(nby, nbx)[:, 1:] & ~dc.reshape(nby, nbx)[:, :-1]]
    assert d.max() <= 5 and abs(v).max() < 64
    # every quarter-pel phase equally often (the headline's tap mix)
    for c in range(4):
        share = np.bincount(v[:, c] & 3, minlength=4) / len(v)
        assert share.min() > 0.2 and share.max() < 0.3, (c, share)
