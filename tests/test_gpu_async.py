"""GPU: round-3 boundary pieces.

* asynchronous transfers (TODO-CUDA:5-7; schrogpuframe.c:480-609 is the synchronous pattern they
  replace): pictures whose coefficients go up from pinned host memory on the H2D queue, are decoded
  on the kernel queues and come down on the D2H queue -- three pictures in flight, ordered by marks
  only -- equal the oracle's; the frame-layer twins schro_frame_to_hip_async / _to_cpu_async;
* references that MOVE between devices: two exec-domain threads and contexts on the one
  device of this box (schro_hip_scheduler_new_on ({0, 0})), a B picture whose references live on
  different "devices": the foreign one reaches it through schro_hip_frame_copy_to (a peer copy of the
  upsampled frame) and the picture equals the oracle's;
* the memory-domain table off an exec-domain thread fails loudly; schro_hip_init;
  schro_upsampled_hipframe_upsample_inplace; schro_hip_codeblock_layout against a Python model."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle_lib as O
import schroedinger_amd as sa
import synth
from schroedinger_amd import _lib, frames

pytestmark = pytest.mark.gpu

W, H, DEPTH, FILT = 320, 192, 3, 0


def picture_inputs(seed):
    P = synth.motion_params(W, H, 12, 8, 2, (1, 1, 1), (1, 1))
    dims = [(H, W), (H // 2, W // 2), (H // 2, W // 2)]
    resid = [synth.image_s(h, w, np.int16, seed=seed + k) for k, (h, w) in enumerate(dims)]
    coeffs = [O.forward_iwt(r, DEPTH, FILT) for r in resid]
    mv = synth.motion_field(P["x_num_blocks"], P["y_num_blocks"], 40, seed=seed + 7)
    return P, dims, coeffs, mv


def test_pictures_through_the_copy_queues(ctx):
    lib = ctx.lib
    P, dims, _, _ = picture_inputs(0)
    refs_np = [[synth.picture_u8(h, w, seed=300 + 10 * r + k) for k, (h, w) in enumerate(dims)] for r in range(2)]
    ups = [[O.UpComp(p) for p in comps] for comps in refs_np]
    hp = [[ctx.hp_plane(h, w) for (h, w) in dims] for _ in range(2)]
    ctx.upsample_batch([(ctx.upload(refs_np[r][k]), hp[r][k]) for r in range(2) for k in range(3)])
    ctx.synchronize()
    npic, slots = 6, 3
    # a slot = one picture in flight: pinned host coefficient planes / MV blob / output planes, device twins
    slot = []
    for s in range(slots):
        slot.append(dict(
            h_co=[ctx.host_array(d, np.int16) for d in dims], h_out=[ctx.host_array(d, np.uint8) for d in dims],
            h_mv=ctx.host_array((1, 20 * P["x_num_blocks"] * P["y_num_blocks"]), np.uint8),
            d_co=[ctx.plane(h, w, np.int16) for (h, w) in dims], d_res=[ctx.plane(h, w, np.int16) for (h, w) in dims],
            d_out=[ctx.plane(h, w, np.uint8) for (h, w) in dims]))
        slot[s]["d_mv"] = ctx.plane(1, slot[s]["h_mv"].shape[1], np.uint8)
    wants, got = [], []
    for n in range(npic + 1):
        if n < npic:
            s = slot[n % slots]
            # the slot's previous picture has come down (its pinned planes are free again)
            if n >= slots:
                ctx.queue_synchronize(ctx.QUEUE_D2H)
                got.append([o.copy() for o in slot[n % slots]["h_out"]])
            _, _, coeffs, mv = picture_inputs(10 * n)
            for k in range(3):
                s["h_co"][k][...] = coeffs[k]
            s["h_mv"][...] = np.ascontiguousarray(mv).view(np.uint8).reshape(1, -1)
            wants.append([O.motion_render(mv, O.MotionParams(**P), k, ups[0][k], ups[1][k],
                                          O.inverse_iwt(coeffs[k], DEPTH, FILT), dims[k][1], dims[k][0]) for k in range(3)])
            ctx.select_queue(ctx.QUEUE_H2D)
            ctx.queue_wait_mark(8 + n % slots)          # the kernels that last read this slot's device planes
            for k in range(3):
                s["d_co"][k].upload_async(s["h_co"][k])
            s["d_mv"].upload_async(s["h_mv"])
            ctx.queue_mark(n % slots)
            ctx.select_queue(n % 2)                      # kernels alternate between the two kernel queues
            ctx.queue_wait_mark(n % slots)
            ctx.queue_wait_mark(12 + n % slots)         # the download that last read this slot's output planes
            ctx.iiwt_batch(list(zip(s["d_co"], s["d_res"])), DEPTH, FILT)
            ctx.obmc_batch([sa.obmc_plane(s["d_mv"], P, k, hp[0][k], hp[1][k], s["d_res"][k], s["d_out"][k])
                            for k in range(3)])
            ctx.queue_mark(8 + n % slots)
            ctx.queue_mark(4 + n % slots)
            ctx.select_queue(ctx.QUEUE_D2H)
            ctx.queue_wait_mark(4 + n % slots)
            for k in range(3):
                s["d_out"][k].download_async(s["h_out"][k])
            ctx.queue_mark(12 + n % slots)
    ctx.select_queue(0)
    ctx.synchronize()
    for n in range(npic - slots, npic):
        got.append([o.copy() for o in slot[n % slots]["h_out"]])
    assert len(got) == npic
    for n in range(npic):
        for k in range(3):
            assert np.array_equal(got[n][k], wants[n][k]), (n, k)


def test_pictures_through_the_copy_queues_handover_order(ctx):
    """The same pipeline in the order DESIGN 5 (r04) prescribes -- the host hands over picture n - 2 (waits for ITS download
    mark and nothing else) before it enqueues picture n; the download of picture n - 1 is enqueued once its kernels' mark has
    fired -- in the combine form (prediction-only OBMC, the wavelet's epilogue writes the picture).  No copy is ever enqueued
    behind an event that has not fired; the pictures are the oracle's."""
    P, dims, _, _ = picture_inputs(0)
    refs_np = [[synth.picture_u8(h, w, seed=300 + 10 * r + k) for k, (h, w) in enumerate(dims)] for r in range(2)]
    ups = [[O.UpComp(p) for p in comps] for comps in refs_np]
    hp = [[ctx.hp_plane(h, w) for (h, w) in dims] for _ in range(2)]
    ctx.upsample_batch([(ctx.upload(refs_np[r][k]), hp[r][k]) for r in range(2) for k in range(3)])
    ctx.synchronize()
    npic, slots = 7, 3
    UP, DONE, DOWN = 0, 4, 12
    slot = []
    for s in range(slots):
        slot.append(dict(
            h_co=[ctx.host_array(d, np.int16) for d in dims], h_out=[ctx.host_array(d, np.uint8) for d in dims],
            h_mv=ctx.host_array((1, 20 * P["x_num_blocks"] * P["y_num_blocks"]), np.uint8),
            d_co=[ctx.plane(h, w, np.int16) for (h, w) in dims], d_pred=[ctx.plane(h, w, np.uint8) for (h, w) in dims],
            d_out=[ctx.plane(h, w, np.uint8) for (h, w) in dims]))
        slot[s]["d_mv"] = ctx.plane(1, slot[s]["h_mv"].shape[1], np.uint8)
    wants, got = [], {}
    for n in range(npic + 2):
        if n >= 2:
            ctx.queue_mark_synchronize(DOWN + (n - 2) % slots)          # picture n - 2 leaves
            got[n - 2] = [o.copy() for o in slot[(n - 2) % slots]["h_out"]]
        if n < npic:
            s = slot[n % slots]
            _, _, coeffs, mv = picture_inputs(10 * n)
            for k in range(3):
                s["h_co"][k][...] = coeffs[k]
            s["h_mv"][...] = np.ascontiguousarray(mv).view(np.uint8).reshape(1, -1)
            wants.append([O.motion_render(mv, O.MotionParams(**P), k, ups[0][k], ups[1][k],
                                          O.inverse_iwt(coeffs[k], DEPTH, FILT), dims[k][1], dims[k][0]) for k in range(3)])
            ctx.select_queue(ctx.QUEUE_H2D)                             # (this queue never waits for anything)
            for k in range(3):
                s["d_co"][k].upload_async(s["h_co"][k])
            s["d_mv"].upload_async(s["h_mv"])
            ctx.queue_mark(UP + n % slots)
            ctx.select_queue(n % 2)
            ctx.queue_wait_mark(UP + n % slots)
            ctx.obmc_batch([sa.obmc_plane(s["d_mv"], P, k, hp[0][k], hp[1][k], None, s["d_pred"][k], prediction_only=True)
                            for k in range(3)])
            ctx.iiwt_batch([(s["d_co"][k], s["d_out"][k], s["d_pred"][k]) for k in range(3)], DEPTH, FILT)
            ctx.queue_mark(DONE + n % slots)
        if 1 <= n <= npic:
            t = slot[(n - 1) % slots]
            ctx.queue_mark_synchronize(DONE + (n - 1) % slots)          # its kernels are done: the copy's event has fired
            ctx.select_queue(ctx.QUEUE_D2H)
            for k in range(3):
                t["d_out"][k].download_async(t["h_out"][k])
            ctx.queue_mark(DOWN + (n - 1) % slots)
    ctx.select_queue(0)
    ctx.synchronize()
    assert sorted(got) == list(range(npic))
    for n in range(npic):
        for k in range(3):
            assert np.array_equal(got[n][k], wants[n][k]), (n, k)


def test_frame_layer_async_twins(ctx):
    lib = ctx.lib
    dims = [(H, W), (H // 2, W // 2), (H // 2, W // 2)]
    src = [ctx.host_array(d, np.int16) for d in dims]
    dst = [ctx.host_array(d, np.int16) for d in dims]
    for k, a in enumerate(src):
        a[...] = synth.image_s(a.shape[0], a.shape[1], np.int16, seed=60 + k)
    fmt = frames.frame_format(np.int16, 1, 1)
    dev = frames.DeviceFrame(ctx, fmt, W, H)
    hsrc, hdst = frames.HostFrame.__new__(frames.HostFrame), None
    hsrc = frames.HostFrame(src, 1, 1)
    # HostFrame copies non-contiguous planes only: the pinned arrays are used as they are
    assert all(p.ctypes.data == a.ctypes.data for p, a in zip(hsrc.planes, src))
    hdst = frames.HostFrame(dst, 1, 1)
    ctx.select_queue(ctx.QUEUE_H2D)
    sa.check(lib.schro_frame_to_hip_async(dev.ptr(), hsrc.ptr()))
    ctx.queue_mark(1)
    ctx.select_queue(ctx.QUEUE_D2H)
    ctx.queue_wait_mark(1)
    sa.check(lib.schro_hipframe_to_cpu_async(hdst.ptr(), dev.ptr()))
    ctx.queue_synchronize(ctx.QUEUE_D2H)
    ctx.select_queue(0)
    for k in range(3):
        assert np.array_equal(dst[k], src[k])
    dev.unref()


def test_reference_moves_between_two_contexts_on_this_device():
    sched = sa.Scheduler(devices=[0, 0])
    assert sched.n_devices == 2
    lib = sched.lib
    P, dims, coeffs, mv = picture_inputs(500)
    fmt8 = frames.frame_format(np.uint8, 1, 1)
    refs_np = {n: [synth.picture_u8(h, w, seed=n + k) for k, (h, w) in enumerate(dims)] for n in (0, 10)}
    keep, result = {}, {}

    def reference(number):
        def run(ctx, dev):
            plain = frames.DeviceFrame(ctx, fmt8, W, H).upload(frames.HostFrame(refs_np[number], 1, 1))
            up = frames.DeviceFrame(ctx, fmt8, W, H, upsampled=True)
            up.c.virt_frame1 = plain.p                     # schrogpuframe.h:29: the one-argument form
            sa.check(lib.schro_upsampled_hipframe_upsample_inplace(up.ptr()))
            sched.publish_reference(dev, up.ptr())
            keep[number] = (plain, up)
            return 0
        return run

    def bipred(ctx, dev):
        f = [C.cast(sched.reference_frame(dev, n), C.POINTER(_lib.Frame)) for n in (10, 0)]
        assert f[0] and f[1]
        d_mv = ctx.upload_bytes(mv)
        res, out = [], []
        for k, (h, w) in enumerate(dims):
            co = ctx.upload(coeffs[k])
            r = ctx.plane(h, w, np.int16)
            ctx.iiwt_batch([(co, r)], DEPTH, FILT)
            res.append(r)
            out.append(ctx.plane(h, w, np.uint8))
        planes = []
        for k in range(3):
            class V:        # a view of the moved frame's component: pointer + band pitch
                pass
            views = []
            for fr in f:
                v = V()
                v.ptr, v.stride = fr.contents.components[k].data, fr.contents.components[k].stride
                v.pair = k > 0 and fr.contents.is_upsampled == 2        # 4:2:0 chroma: one (U, V) pair image
                views.append(v)
            planes.append(sa.obmc_plane(d_mv, P, k, views[0], views[1], res[k], out[k]))
        ctx.obmc_batch(planes)
        result["B"] = [o.download() for o in out]
        return 0

    d0, _ = sched.submit(0, [], True, reference(0))
    d1, _ = sched.submit(10, [], True, reference(10))
    assert d0 != d1
    dev, foreign = sched.submit(11, [10, 0], False, bipred)
    assert dev == d1 and foreign == 0
    sched.retire(0)                 # retired while the dependent may still be queued
    sched.retire(10)
    assert sched.wait() == 0
    assert sched.moves() == 1
    ups = {n: [O.UpComp(p) for p in refs_np[n]] for n in (0, 10)}
    for k, (h, w) in enumerate(dims):
        want = O.motion_render(mv, O.MotionParams(**P), k, ups[10][k], ups[0][k], O.inverse_iwt(coeffs[k], DEPTH, FILT), w, h)
        assert np.array_equal(result["B"][k], want), k
    sched.close()


def test_two_reference_pictures_of_one_device_in_flight():
    """r04 (VERDICT r03 missing 2): the scheduler drains nothing.  A chain of reference pictures on one
    device -- each a 2160p upsample + an inverse wavelet of distinct buffers, ~0.1 ms of device work against
    a few tens of microseconds of enqueueing -- leaves several of them in flight at once
    (schro_hip_scheduler_refs_in_flight_max, from the pictures' `ready` events); the last picture, on the
    other context, predicts across the chains from frames that were never waited for on the host, and
    equals the oracle's."""
    sched = sa.Scheduler(devices=[0, 0])
    lib = sched.lib
    P, dims, coeffs, mv = picture_inputs(900)
    fmt8 = frames.frame_format(np.uint8, 1, 1)
    refs_np = {n: [synth.picture_u8(h, w, seed=3 * n + k) for k, (h, w) in enumerate(dims)] for n in range(6)}
    keep, result = {}, {}
    # everything the reference pictures need is allocated and uploaded BEFORE they are submitted (uploads and
    # allocations wait for the device): their functions only enqueue kernels.  Chain 0 - 3 goes to the first
    # context (least loaded, ties to the lowest index), 4 - 5 to the second.
    big, plains, ups_dev = {}, {}, {}
    for dev, numbers in ((0, (0, 1, 2, 3)), (1, (4, 5))):
        ctx = sched.contexts[dev]
        big[dev] = (ctx.upload(synth.picture_u8(2160, 3840, seed=5)), [ctx.hp_plane(2160, 3840) for _ in range(2)])
        for n in numbers:
            plains[n] = frames.DeviceFrame(ctx, fmt8, W, H).upload(frames.HostFrame(refs_np[n], 1, 1))
            ups_dev[n] = frames.DeviceFrame(ctx, fmt8, W, H, upsampled=True)
        ctx.synchronize()

    gate = threading.Event()        # holds picture 0 until everything is submitted: the second chain finds device 0 loaded

    def reference(number):
        def run(ctx, dev):
            if number == 0:
                gate.wait(60)
            src, hps = big[dev]
            for hp in hps:                              # device work that outlasts the host's enqueueing
                ctx.upsample_batch([(src, hp)])
            plain, up = plains[number], ups_dev[number]
            # (the frame layer's upsample would wait for its own completion: the plane layer, as a pipelined host does)
            c, pc = up.c.components, plain.c.components
            planes = (_lib.UpsamplePlane * 2)()
            planes[0] = _lib.UpsamplePlane(pc[0].data, pc[0].stride, c[0].data, c[0].stride, pc[0].width, pc[0].height, None, 0)
            planes[1] = _lib.UpsamplePlane(pc[1].data, pc[1].stride, c[1].data, c[1].stride, pc[1].width, pc[1].height,
                                           pc[2].data, pc[2].stride)
            sa.check(lib.schro_hip_upsample_batch(ctx.h, planes, 2))
            up.c.upsample_done = 1
            for _ in range(4):
                for hp in hps:
                    ctx.upsample_batch([(src, hp)])
            sched.publish_reference(dev, up.ptr())
            keep[number] = (plain, up)
            return 0
        return run

    def bipred(ctx, dev):
        f = [C.cast(sched.reference_frame(dev, n), C.POINTER(_lib.Frame)) for n in (5, 3)]
        assert f[0] and f[1]
        d_mv = ctx.upload_bytes(mv)
        res, out = [], []
        for k, (h, w) in enumerate(dims):
            co = ctx.upload(coeffs[k])
            r = ctx.plane(h, w, np.int16)
            ctx.iiwt_batch([(co, r)], DEPTH, FILT)
            res.append(r)
            out.append(ctx.plane(h, w, np.uint8))
        planes = []
        for k in range(3):
            class V:
                pass
            views = []
            for fr in f:
                v = V()
                v.ptr, v.stride = fr.contents.components[k].data, fr.contents.components[k].stride
                v.pair = k > 0 and fr.contents.is_upsampled == 2
                views.append(v)
            planes.append(sa.obmc_plane(d_mv, P, k, views[0], views[1], res[k], out[k]))
        ctx.obmc_batch(planes)
        result["B"] = [o.download() for o in out]
        return 0

    d0, _ = sched.submit(0, [], True, reference(0))
    for n in (1, 2, 3):                                 # a chain of references on d0, nothing waits in between
        dev, _ = sched.submit(n, [n - 1], True, reference(n))
        assert dev == d0
    d1, _ = sched.submit(4, [], True, reference(4))
    assert d1 != d0
    dev, _ = sched.submit(5, [4], True, reference(5))
    assert dev == d1
    dev, foreign = sched.submit(6, [5, 3], False, bipred)
    assert dev == d1 and foreign == 3
    gate.set()
    assert sched.wait() == 0
    assert sched.moves() == 1 and sched.skipped() == 0
    assert sched.refs_in_flight_max() >= 2, sched.refs_in_flight_max()
    ups = {n: [O.UpComp(p) for p in refs_np[n]] for n in (5, 3)}
    for k, (h, w) in enumerate(dims):
        want = O.motion_render(mv, O.MotionParams(**P), k, ups[5][k], ups[3][k], O.inverse_iwt(coeffs[k], DEPTH, FILT), w, h)
        assert np.array_equal(result["B"][k], want), k
    sched.close()


def test_retired_references_then_more_on_the_same_device_and_the_second_kernel_queue():
    """r05.  (1) A reference is submitted, RETIRED and complete -- its record and `ready` event are
    gone -- before the next reference picture of the same device runs: the in-flight statistic must not look at
    the destroyed event (it owns events of its own now).  (2) A dependent that selects the SECOND kernel queue
    follows the reference all the same: the reference's upsample sits behind a pile of device work on queue 0,
    and the prediction from it, enqueued on queue 1 without any host wait, equals the oracle's.  (3) Every
    picture function starts with queue 0 selected, whatever the one before it left selected.  (4) Frames the
    scheduler lets go of are released behind the device's kernel queues: the run ends with nothing pending."""
    sched = sa.Scheduler(devices=[0])
    lib = sched.lib
    P, dims, coeffs, mv = picture_inputs(1300)
    fmt8 = frames.frame_format(np.uint8, 1, 1)
    refs_np = {n: [synth.picture_u8(h, w, seed=5 * n + k) for k, (h, w) in enumerate(dims)] for n in range(4)}
    ctx0 = sched.contexts[0]
    big = (ctx0.upload(synth.picture_u8(2160, 3840, seed=9)), [ctx0.hp_plane(2160, 3840) for _ in range(2)])
    pre = {}
    for n in range(4):
        pre[n] = (frames.DeviceFrame(ctx0, fmt8, W, H).upload(frames.HostFrame(refs_np[n], 1, 1)),
                  frames.DeviceFrame(ctx0, fmt8, W, H, upsampled=True))
    d_mv = ctx0.upload_bytes(mv)
    d_co = [ctx0.upload(coeffs[k]) for k in range(3)]
    d_res = [ctx0.plane(h, w, np.int16) for (h, w) in dims]
    outs = {n: [ctx0.plane(h, w, np.uint8) for (h, w) in dims] for n in (10, 11)}
    ctx0.synchronize()
    keep, seen_queue = {}, []

    def reference(number, pile):
        def run(ctx, dev):
            seen_queue.append(ctx.queue())
            src, hps = big
            for _ in range(pile):                       # device work in front of the reference's own
                for hp in hps:
                    ctx.upsample_batch([(src, hp)])
            plain, up = pre[number]
            c, pc = up.c.components, plain.c.components
            planes = (_lib.UpsamplePlane * 2)()
            planes[0] = _lib.UpsamplePlane(pc[0].data, pc[0].stride, c[0].data, c[0].stride, pc[0].width, pc[0].height, None, 0)
            planes[1] = _lib.UpsamplePlane(pc[1].data, pc[1].stride, c[1].data, c[1].stride, pc[1].width, pc[1].height,
                                           pc[2].data, pc[2].stride)
            sa.check(lib.schro_hip_upsample_batch(ctx.h, planes, 2))
            up.c.upsample_done = 1
            sched.publish_reference(dev, up.ptr())
            keep[number] = (plain, up)
            return 0
        return run

    def dependent(number, refs):
        def run(ctx, dev):
            seen_queue.append(ctx.queue())
            ctx.select_queue(1)                         # the header's second kernel queue; left selected on purpose
            f = [C.cast(sched.reference_frame(dev, n), C.POINTER(_lib.Frame)) for n in refs]
            assert f[0] and f[1]
            if number == 10:
                ctx.iiwt_batch(list(zip(d_co, d_res)), DEPTH, FILT)
            planes = []
            for k in range(3):
                class V:
                    pass
                views = []
                for fr in f:
                    v = V()
                    v.ptr, v.stride = fr.contents.components[k].data, fr.contents.components[k].stride
                    v.pair = k > 0 and fr.contents.is_upsampled == 2
                    views.append(v)
                planes.append(sa.obmc_plane(d_mv, P, k, views[0], views[1], d_res[k], outs[number][k]))
            ctx.obmc_batch(planes)
            return 0
        return run

    # reference 0: submitted, retired, complete -> its record (and event) go before reference 1 runs
    sched.submit(0, [], True, reference(0, 1))
    sched.retire(0)
    assert sched.wait() == 0
    sched.submit(1, [], True, reference(1, 12))
    sched.submit(2, [], True, reference(2, 12))
    sched.submit(10, [1, 2], False, dependent(10, (1, 2)))
    sched.retire(1)
    sched.submit(3, [2], True, reference(3, 12))
    sched.submit(11, [3, 2], False, dependent(11, (3, 2)))
    sched.retire(2)
    sched.retire(3)
    assert sched.wait() == 0
    assert seen_queue == [0] * len(seen_queue), seen_queue
    ctx0.synchronize()
    resid = [O.inverse_iwt(coeffs[k], DEPTH, FILT) for k in range(3)]
    ups = {n: [O.UpComp(p) for p in refs_np[n]] for n in (1, 2, 3)}
    for number, (a, b) in ((10, (1, 2)), (11, (3, 2))):
        for k, (h, w) in enumerate(dims):
            want = O.motion_render(mv, O.MotionParams(**P), k, ups[a][k], ups[b][k], resid[k], w, h)
            assert np.array_equal(outs[number][k].download(), want), (number, k)
    sched.close()


@pytest.mark.timeout(780)
def test_bench_two_ranks_on_this_device():
    """The N > 1 path of bench.py (launcher, rank processes, gloo rendezvous, max over ranks) on the one GPU
    of this box: two ranks share the device (SCHRO_BENCH_SHARE_DEVICE=1), rc 0, one JSON line, n_gpus 2."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCHRO_BENCH_SHARE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    # (on a fresh box the first `import torch` pages the libraries in for a minute or two: once, here, not twice at the
    # same time in the two ranks with the other one waiting at the rendezvous)
    subprocess.run([sys.executable, "-c", "import torch, torch.distributed"], env=env, capture_output=True, timeout=400)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--headline-only", "--prewarm-ms", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0 and out["scaling"] == "weak"


def test_domain_table_off_the_exec_domain_thread(ctx):
    lib = ctx.lib
    lib.schro_hip_init()
    dom = C.cast(lib.schro_hip_context_domain(ctx.h), C.POINTER(_lib.MemoryDomain)).contents
    lib.schro_hip_thread_bind(ctx.h)
    p = dom.alloc(4096)
    assert p
    dom.free(p, 4096)
    out = {}

    def other():                     # a thread that never bound a domain: no silent device 0
        out["p"] = dom.alloc(4096)
        out["err"] = lib.schro_hip_last_error().decode()
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert not out["p"] and "exec-domain thread" in out["err"]


def test_codeblock_layout_matches_the_decoders_geometry(ctx):
    w, h, depth, stride = 360, 208, 3, 768
    hc, vc = [1, 3, 4, 5], [1, 2, 3, 4]
    tab = ctx.codeblock_layout(w, h, depth, hc, vc, stride, 2)
    want = []
    for index in range(1 + 3 * depth):
        position = 0 if index == 0 else (((index - 1) // 3) << 2) | ((index - 1) % 3 + 1)
        level = position >> 2
        shift = depth - level
        bw, bh, bstride = w >> shift, h >> shift, stride << shift
        base = (bstride >> 1 if position & 2 else 0) + (bw * 2 if position & 1 else 0)
        nh, nv = (hc[0], vc[0]) if position == 0 else (hc[level + 1], vc[level + 1])
        for y in range(nv):
            y0, y1 = bh * y // nv, bh * (y + 1) // nv
            # schrodecoder.c:3565-3577: widths by an error accumulator
            xmin, acc, cw = 0, 0, bw // nh
            inc = bw - nh * cw
            for x in range(nh):
                x0 = xmin
                xmin += cw
                acc += inc
                if acc >= nh:
                    acc -= nh
                    xmin += 1
                want.append((base + y0 * bstride + 2 * x0, bstride, xmin - x0, y1 - y0, -1))
    assert len(tab) == len(want)
    for t, wv in zip(tab, want):
        assert (t.dst_offset, t.dst_stride, t.width, t.height, t.src_offset) == wv
