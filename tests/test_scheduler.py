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


def test_retire_before_the_dependent_runs():
    """schro_decoder_reference_retire runs at PARSE time (schrodecoder.c:1302): references are retired
    while pictures that predict from them -- on another device, so they wait -- are still queued, or
    while the reference itself is.  Round 2 deadlocked here: completion was looked up by
    number, and retire had erased the number."""
    s = sa.Scheduler(2, virtual=True)
    gate = threading.Event()
    order, lock = [], threading.Lock()

    def f(name, wait=False):
        def g(ctx, index):
            if wait:
                gate.wait(5)
            with lock:
                order.append(name)
            return 0
        return g
    d0, _ = s.submit(0, [], True, f("I0", wait=True))       # held back
    d1, _ = s.submit(10, [], True, f("I10"))
    assert d0 != d1
    dev, foreign = s.submit(12, [10, 0], False, f("B12"))    # waits for picture 0 on the other device
    assert dev == d1 and foreign == 0
    s.retire(0)                                              # ... which is retired before it has even run
    s.retire(10)
    with pytest.raises(sa.SchroHipError):
        s.submit(13, [0], False, f("late"))                  # retired: new pictures cannot predict from it
    done = threading.Event()
    result = []

    def waiter():
        result.append(s.wait())
        done.set()
    threading.Thread(target=waiter, daemon=True).start()
    time.sleep(0.05)
    assert not done.is_set()                                 # B12 still waits for I0
    gate.set()
    assert done.wait(5), "scheduler_wait hangs after retire (the round-2 deadlock)"
    assert result == [0]
    assert order.index("I0") < order.index("B12") and order.index("I10") < order.index("B12")
    s.close()


def test_references_move_to_the_dependents_device():
    """A prediction across two chains: the foreign reference's published frame is brought to the
    picture's device before its function runs, once per device, and handed out by number."""
    s = sa.Scheduler(2, virtual=True)
    seen = {}
    gate = threading.Event()        # (holds the first anchor until the second has been placed: "least loaded" then sees its device busy)

    def ref(token, wait=False):
        def g(ctx, index):
            if wait:
                gate.wait(5)
            s.publish_reference(index, token)
            return 0
        return g

    def dep(name, numbers):
        def g(ctx, index):
            seen[name] = (index, [s.reference_frame(index, n) for n in numbers])
            return 0
        return g
    d0, _ = s.submit(0, [], True, ref(0x1000, wait=True))
    d1, _ = s.submit(10, [], True, ref(0x2000))
    gate.set()
    assert d0 != d1
    dev, foreign = s.submit(11, [10, 0], False, dep("B11", [10, 0]))
    assert (dev, foreign) == (d1, 0)
    dev2, _ = s.submit(12, [10, 0], False, dep("B12", [10, 0]))          # the copy is there already
    dev3, foreign3 = s.submit(1, [0, 10], False, dep("B1", [0, 10]))       # and the other way round
    assert dev3 == d0 and foreign3 == 10
    s.retire(0)
    s.retire(10)
    assert s.wait() == 0
    assert seen["B11"] == (d1, [0x2000, 0x1000]) and seen["B12"] == (d1, [0x2000, 0x1000])
    assert seen["B1"] == (d0, [0x1000, 0x2000])
    assert s.moves() == 2                                    # picture 0 -> d1 once, picture 10 -> d0 once
    s.close()


def test_a_reused_picture_number_is_a_new_reference():
    s = sa.Scheduler(1, virtual=True)
    got = []
    s.submit(5, [], True, lambda c, i: s.publish_reference(i, 0x51) or 0)
    s.submit(6, [5], False, lambda c, i: got.append(s.reference_frame(i, 5)) or 0)
    s.submit(5, [], True, lambda c, i: s.publish_reference(i, 0x52) or 0)       # the number comes round again
    s.submit(7, [5], False, lambda c, i: got.append(s.reference_frame(i, 5)) or 0)
    assert s.wait() == 0
    assert got == [0x51, 0x52]
    s.close()


def test_a_failed_reference_skips_its_dependents_everywhere():
    """r04: a reference whose function fails is complete -- nobody waits for it for
    ever -- but FAILED: the pictures that predict from it, on its device and on others, and the pictures
    that predict from those, do not run (SCHRO_HIP_ESKIPPED), as the reference decoder skips pictures whose
    parse hit an error (picture->error, schrodecoder.c:1308-1311, :1399-1418).  Chains that do not touch the
    failed picture are unaffected, and wait () reports the real error once."""
    s = sa.Scheduler(2, virtual=True)
    ran = []
    lock = threading.Lock()

    def ok(number):
        def f(ctx, index):
            with lock:
                ran.append(number)
            return 0
        return f

    gate = threading.Event()        # (holds the anchor until chain B has been placed: "least loaded" then sees device d0 busy)

    def broken(ctx, index):
        gate.wait(30)
        with lock:
            ran.append("broken")
        return -2                                       # SCHRO_HIP_EDEVICE

    d0, _ = s.submit(0, [], True, broken)               # chain A's anchor fails
    d1, _ = s.submit(100, [], True, ok(100))            # chain B is healthy
    gate.set()
    assert d0 != d1
    s.submit(1, [0], True, ok(1))                       # P from the failed anchor: skipped, and fails in turn
    s.submit(2, [0, 1], False, ok(2))                   # B between them: skipped
    s.submit(3, [1], True, ok(3))                       # second generation: skipped
    s.submit(101, [100], True, ok(101))
    dev, foreign = s.submit(102, [100, 0], False, ok(102))      # across the chains, from the failed anchor: skipped
    assert dev == d1 and foreign == 0
    s.submit(103, [100, 101], False, ok(103))
    assert s.wait() == -2
    assert sorted(x for x in ran if x != "broken") == [100, 101, 103]
    assert s.skipped() == 4
    assert s.moves() == 0                               # nothing is moved for a picture that does not run
    # a later, healthy chain on the same devices runs
    s.submit(200, [], True, ok(200))
    s.submit(201, [200], False, ok(201))
    assert s.wait() == 0                                # (the error was reported once)
    assert 201 in ran and s.skipped() == 4
    s.close()


def test_both_references_the_same_foreign_picture():
    """refs = {n, n} is legal in Dirac: the foreign frame is moved ONCE (two copies were made and
    the first leaked)."""
    s = sa.Scheduler(2, virtual=True)
    seen = {}

    def ref(number):
        def f(ctx, index):
            s.publish_reference(index, 1000 + number)
            return 0
        return f

    def twice(ctx, index):
        seen["frames"] = (s.reference_frame(index, 7), s.reference_frame(index, 7))
        return 0

    d0, _ = s.submit(7, [], True, ref(7))
    d1, _ = s.submit(50, [], True, ref(50))
    assert d0 != d1
    dev, foreign = s.submit(51, [50, 7], False, lambda c, i: 0)       # brings 7 over to d1 (one move)
    assert dev == d1 and foreign == 7
    s.wait()
    assert s.moves() == 1
    # a picture on d0's side whose two references are both picture 50
    dev, foreign = s.submit(8, [7, 50], True, lambda c, i: 0)         # one more move: 50 -> d0
    assert dev == d0
    s.wait()
    assert s.moves() == 2
    dev, _ = s.submit(52, [50, 50], False, lambda c, i: 0)            # at home: nothing to move
    s.wait()
    assert s.moves() == 2
    s.submit(9, [7], True, lambda c, i: 0)
    s.retire(50)
    s.wait()
    s.close()


def test_a_new_chain_goes_to_the_device_with_fewer_live_references():
    """r05: two anchors in a row with workers fast enough that both devices are idle at every submit -- the second
    chain still goes to the other device (its dependents are still to come), and a retired reference does not count."""
    s = sa.Scheduler(2, virtual=True)
    d0, _ = s.submit(0, [], True, lambda c, i: 0)
    assert s.wait() == 0                                     # nothing outstanding anywhere
    d1, _ = s.submit(10, [], True, lambda c, i: 0)
    assert s.wait() == 0
    assert d0 != d1
    s.retire(0)
    d2, _ = s.submit(20, [], True, lambda c, i: 0)           # device d0 holds no live reference any more
    assert s.wait() == 0
    assert d2 == d0
    s.close()
