"""CPU: the multi-device scheduler's host logic (schro_hip_scheduler_*) on
virtual devices: reference-chain affinity, references before dependents, chains spread over the
devices, predictions across chains reported and ordered.  (The same object with real devices runs
in tests/test_gpu_scheduler.py.)"""
import threading
import time

import pytest

import schroedinger_amd as sa


def gop_stream(n_gops, gop_len):
    """Coded-order pictures of closed GOPs: I, then P pictures each predicting from the previous
    anchor, B pictures between two anchors (non-reference) -- the shape of test_stream.drc."""
    pics, num = [], 0
    for g in range(n_gops):
        anchor = num
        pics.append((num, [], True))
        num += 1
        while num < (g + 1) * gop_len:
            p = num
            pics.append((p, [anchor], True))            # P
            if p + 1 < (g + 1) * gop_len:
                pics.append((p + 1, [anchor, p], False))  # B between the two anchors
                num += 2
            else:
                num += 1
            anchor = p
    return pics


@pytest.mark.parametrize("ndev", [1, 2, 4])
def test_chains_stay_on_a_device_and_references_come_first(ndev):
    s = sa.Scheduler(ndev, virtual=True)
    assert s.n_devices == ndev
    lock = threading.Lock()
    finished, ran_on, violations = set(), {}, []
    pics = gop_stream(8, 9)

    def body(number, refs):
        def f(ctx, index):
            assert ctx is None                           # virtual device
            with lock:
                missing = [r for r in refs if r not in finished]
                if missing:
                    violations.append((number, missing))
            time.sleep(0.001)
            with lock:
                finished.add(number)
                ran_on[number] = index
            return 0
        return f

    owner = {}
    for number, refs, is_ref in pics:
        dev, foreign = s.submit(number, refs, is_ref, body(number, refs))
        assert foreign == -1
        owner[number] = dev
        if refs:
            assert dev == owner[refs[0]]                 # follows its first reference
    assert s.wait() == 0
    assert not violations
    assert ran_on == owner and len(finished) == len(pics)
    # 8 closed GOPs over the devices: every device got work, chains never split
    assert len(set(owner.values())) == ndev
    s.close()


def test_prediction_across_chains_waits_for_the_foreign_reference():
    s = sa.Scheduler(2, virtual=True)
    order, lock = [], threading.Lock()
    gate = threading.Event()

    def f(name, wait_gate=False):
        def g(ctx, index):
            if wait_gate:
                gate.wait(5)
            with lock:
                order.append((name, index))
            return 0
        return g
    d0, _ = s.submit(0, [], True, f("I0", wait_gate=True))     # chain A, held back
    d1, _ = s.submit(10, [], True, f("I10"))                   # chain B on the other device
    assert d0 != d1
    dev, foreign = s.submit(11, [10, 0], False, f("B11"))      # predicts from both chains
    assert dev == d1 and foreign == 0
    time.sleep(0.05)
    assert ("B11", d1) not in order                            # still waiting for picture 0
    gate.set()
    assert s.wait() == 0
    assert order.index(("I0", d0)) < order.index(("B11", d1))
    s.close()


def test_errors_and_retirement():
    s = sa.Scheduler(2, virtual=True)
    with pytest.raises(sa.SchroHipError, match="never submitted"):
        s.submit(5, [4], False, lambda c, i: 0)
    s.submit(1, [], True, lambda c, i: 7)                      # a failing picture function
    assert s.wait() == 7
    s.retire(1)
    with pytest.raises(sa.SchroHipError):
        s.submit(2, [1], False, lambda c, i: 0)                # retired: no longer a reference
    s.close()
    with pytest.raises(sa.SchroHipError):
        sa.Scheduler(0, virtual=False) if sa.device_count() == 0 else (_ for _ in ()).throw(sa.SchroHipError("skip"))
