// scheduler.cpp -- one decode loop, N devices.
//
// The reference schedules picture STAGES on a pool of worker threads and, for its GPU back end,
// on ONE extra exec-domain thread bound to the device (schro_async_add_exec_domain,
// schroasync-pthread.c:362-390; schro_decoder_async_schedule, schrodecoder.c:1546-1682 picks a
// picture whose next stage may run in the calling thread's exec domain).  With several devices
// the missing piece is AFFINITY: every stage of a picture that touches its references
// (x_upsample, x_render_motion, x_combine) must run where those references live.  This is that
// piece behind a C ABI:
//
//   * one exec-domain thread per device, each with its own SchroHipContext (queues, memory
//     domain, caches) -- the reference's SchroThread with exec_domain = the device;
//   * schro_picture_new's decision (schrodecoder.c:332-400: which domain the picture's frames
//     are allocated in) becomes schro_hip_scheduler_submit: a picture that predicts goes to
//     the device of its first reference, so a reference chain (a closed GOP) stays on one
//     device and no reference ever crosses xGMI; a picture without references (an intra
//     picture, every VC-2 low-delay picture) starts a chain on the least loaded device (ties: the one with the fewest live references);
//   * a device runs its pictures in submission order -- coded order, in which references
//     precede the pictures that use them -- so the wavelet-before-render and
//     reference-before-dependent orderings of schrodecoder.c:1589-1660 hold by construction;
//   * r03: a picture whose references ended up on two devices (a prediction across chains) waits
//     for the foreign reference and then MOVES it: the frame its owner published
//     (schro_hip_scheduler_publish_reference) is copied to this device with one
//     hipMemcpyPeerAsync per component before the picture's function runs, and the function asks
//     for "the frame of reference n on my device" (schro_hip_scheduler_reference_frame).
// No data-path collective, no RCCL: pictures shard.
//
// r04 -- events, not drains (TODO-CUDA:5-7).  A picture's function only ENQUEUES on its context's queues.
// When a reference picture's function has returned, the worker joins the context's queues behind ONE
// event (`ready`) and the reference is complete: pictures of the same device follow it in the in-order
// queues, a picture on another device makes its copy queue WAIT for the event (hipStreamWaitEvent), issues
// the peer copy asynchronously and makes its kernel queues wait for the copy -- the scheduler itself waits
// for no device on the path, so a device can have any number of reference pictures in flight
// (refs_in_flight_max counts them from events of its own).  One caveat: on ROCm 7.2 an asynchronous copy
// from / to pinned HOST memory enqueued behind an event that has not fired returns to its caller only when the
// event has.  The PEER copy of a reference does not: measured under rocprofv3 --hip-trace with two contexts on one
// device, the worker spends 22 us in hipMemcpyPeerAsync while 7 ms of the producer's work are outstanding
// (profiles/r05_peer_copy_hip_trace.txt; a copy between two real devices is unmeasured -- were it to hold the
// caller, it would hold the worker of the device that NEEDS the frame, which runs its pictures in coded order
// anyway).  r03 drained the whole device after every reference picture and copied synchronously (the
// reference's precedent: schrogpuframe.c:480-609).
// r05: (1) after `ready` is recorded on queue 0 the context's second kernel queue waits for it as
// well, and every picture function starts with queue 0 selected -- so a dependent on the same device follows
// the reference whichever KERNEL queue it uses (copy queues are ordered by the function's own marks, as in
// a function that uploads into a buffer an earlier picture reads must wait for that picture's
// mark); (2) the in-flight statistic owns its events (the records' `ready` events are destroyed with the
// records); (3) a frame leaves the scheduler -- and its memory becomes reusable -- only when the kernel
// queues of its device have passed the point of its release: dependents whose functions have returned may
// still be reading it, and the next peer copy into the recycled slot runs on a copy queue that is not
// ordered against them.
// A reference whose function FAILED is complete too (nobody waits forever) but marked failed: its
// dependents -- on this device and on others -- do not run their functions, they finish with
// SCHRO_HIP_ESKIPPED, and a skipped reference fails in turn; the reference decoder does the same with
// picture->error / picture->skip (schrodecoder.c:1308-1311, :1399-1418).
//
// Lifetime (r03).  The reference decoder retires a reference at PARSE time
// (schro_decoder_reference_retire, schrodecoder.c:1302), i.e. possibly before pictures that
// predict from it -- already submitted -- have run, and even before the reference itself has.  So
// a reference's state is not looked up by number when a dependent runs: submit resolves the
// numbers to state records once, a record counts the submitted pictures that still need it, and
// retire only removes the number from the lookup table; the record (and the frames it holds) goes
// when it is retired, complete and no longer needed.  Frames are released on the thread of the
// device whose context owns them (a context is not thread-safe).
#include "schro_hip_internal.h"

#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

namespace schro {
// context.cpp: queue 0 of the context waits for its other queues, then `ev` is recorded on it
int context_join_queues (SchroHipContext * ctx, hipEvent_t ev);
// frame.cpp: a copy of `src` on dst_ctx's device, enqueued on its host-to-device copy queue behind `wait_for`;
// `done` is recorded behind the copy and dst_ctx's kernel queues wait for it.  Nothing is waited for (measured: 22 us
// in the call behind an unfired `wait_for`, see the note on r04 above).
SchroHipFrame *frame_copy_to_async (SchroHipContext * dst_ctx, SchroHipFrame * src, hipEvent_t wait_for, hipEvent_t done);
// context.cpp: a context that does not become the calling thread's domain
SchroHipContext *context_new_unbound (int device);
}

struct SchroHipScheduler {
  struct Ref {
    int number;
    int device;                 // index of the owner
    bool complete = false;      // the owner's function has returned and `ready` is recorded behind its device work
    bool failed = false;        // ... with an error (or was skipped): dependents are skipped
    bool retired = false;       // no longer in the lookup table
    hipEvent_t ready = nullptr; // real devices: behind everything the owner's function enqueued
    std::vector < hipEvent_t > copied;  // behind the peer copies made FROM its frame (waited for before the frame goes)
    int users = 0;              // submitted pictures that predict from it and have not finished
    void *frame = nullptr;      // what the owner published (a SchroHipFrame * on real devices)
    std::map < int, void * >copies;       // device index -> the frame moved there
  };
  struct Task {
    int number;
    Ref *self;                  // the picture's own record when it is a reference
    std::vector < Ref * >refs;
    SchroHipPictureFunc func;
    void *priv;
  };
  struct Device {
    SchroHipContext *ctx = nullptr;
    int device = -1;
    std::thread thread;
    std::deque < Task > queue;
    long submitted = 0, finished = 0;
    const Task *current = nullptr;      // the task whose function is running
    // frames of this device's context to release on its thread, and the events to wait for before
    struct Garbage {
      SchroHipFrame *frame;
      std::vector < hipEvent_t > after;
      hipEvent_t destroy;
    };
    std::vector < Garbage > garbage;
    // garbage whose frame still waits for this device's kernel queues to pass the point of release
    // (`after` = the events recorded there); polled by the worker, never waited for while work is runnable
    std::vector < Garbage > pending;
    // events of the statistic's own, one behind each reference picture of this device, oldest first
    // (NOT the records' `ready` events: those are destroyed with their records)
    std::deque < hipEvent_t > in_flight;
  };
  std::vector < Device > devs;
  std::mutex mutex;
  std::condition_variable work, done;
  std::map < int, Ref * >owner; // picture number -> record (references not yet retired)
  bool quit = false;
  bool virtual_devices = false;
  int first_error = 0;
  long moves = 0, skipped = 0;
  int refs_in_flight_max = 0;   // most reference pictures of one device whose device work was still running
};

namespace {

typedef SchroHipScheduler::Ref Ref;

// (mutex held) the record has no further use: hand its frames to the threads that own them
void
release_if_unused (SchroHipScheduler * s, Ref * r)
{
  if (!r->retired || !r->complete || r->users > 0)
    return;
  if (!s->virtual_devices) {
    // the published frame goes when the copies made from it have run; `ready` is destroyed with it
    s->devs[r->device].garbage.push_back ({ (SchroHipFrame *) r->frame, r->copied, r->ready });
    for (auto & c:r->copies)
      s->devs[c.first].garbage.push_back ({ (SchroHipFrame *) c.second, {}, nullptr });
  }
  delete r;
}

// On the thread of the device that owns the frames.  A frame is not released here: the peer copies made from
// it must have run (`after`, recorded by other devices' workers) AND this device's kernel queues must have
// passed this point -- kernels of dependents whose functions have returned may still read it, and once the
// slot is recycled the next writer may be a copy queue that is not ordered against them.  So two
// more events go behind the kernel queues and the entry moves to `pending`.
void
empty_garbage (SchroHipContext * ctx, std::vector < SchroHipScheduler::Device::Garbage > &g,
    std::vector < SchroHipScheduler::Device::Garbage > &pending)
{
  for (auto & e:g) {
    if (e.frame && ctx) {
      for (int q = 0; q < 2; q++) {
        hipEvent_t ev = nullptr;
        hipStream_t st = ctx->streams[q];
        if (st && hipEventCreateWithFlags (&ev, hipEventDisableTiming) == hipSuccess) {
          if (hipEventRecord (ev, st) == hipSuccess)
            e.after.push_back (ev);
          else
            (void) hipEventDestroy (ev);
        }
      }
    }
    pending.push_back (e);
  }
  g.clear ();
}

// release what has become releasable; `wait`: block for the rest (shutdown).  Returns whether entries are left.
bool
poll_pending (std::vector < SchroHipScheduler::Device::Garbage > &pending, bool wait)
{
  size_t keep = 0;
  for (size_t k = 0; k < pending.size (); k++) {
    auto & e = pending[k];
    bool fired = true;
    for (hipEvent_t ev:e.after) {
      if (wait)
        (void) hipEventSynchronize (ev);
      else if (hipEventQuery (ev) == hipErrorNotReady) {
        fired = false;
        break;
      }
    }
    if (!fired) {
      if (keep != k)
        pending[keep] = e;
      keep++;
      continue;
    }
    for (hipEvent_t ev:e.after)
      (void) hipEventDestroy (ev);
    if (e.frame)
      schro_hip_frame_unref (e.frame);
    if (e.destroy)
      (void) hipEventDestroy (e.destroy);
  }
  pending.resize (keep);
  return keep != 0;
}

bool
runnable (const SchroHipScheduler * s, const SchroHipScheduler::Device & d, int index)
{
  if (d.queue.empty ())
    return false;
  for (const Ref * r:d.queue.front ().refs)
    if (r->device != index && !r->complete)
      return false;             // a foreign reference that is not there yet
  return true;
}

void
worker (SchroHipScheduler * s, int index)
{
  SchroHipScheduler::Device & d = s->devs[index];
  if (d.ctx) {
    (void) hipSetDevice (d.device);     // the exec domain of this thread
    schro_hip_thread_bind (d.ctx);      // ... and the memory domain its alloc / free table serves
  }
  std::unique_lock < std::mutex > lock (s->mutex);
  for (;;) {
    auto awake =[&] {
      return s->quit || runnable (s, d, index) || !d.garbage.empty ();
    };
    if (d.pending.empty ())
      s->work.wait (lock, awake);
    else                        // frames waiting for this device's queues: look again in a moment
      s->work.wait_for (lock, std::chrono::milliseconds (1), awake);
    if (!d.garbage.empty () || !d.pending.empty ()) {
      std::vector < SchroHipScheduler::Device::Garbage > g;
      g.swap (d.garbage);
      lock.unlock ();
      empty_garbage (d.ctx, g, d.pending);     // (`pending` is this thread's alone while it runs)
      poll_pending (d.pending, false);
      lock.lock ();
      if (!d.garbage.empty ())
        continue;
    }
    if (!runnable (s, d, index)) {
      if (s->quit && d.garbage.empty ()) {
        lock.unlock ();
        poll_pending (d.pending, true);
        lock.lock ();
        if (d.garbage.empty ())
          return;
      }
      continue;
    }
    SchroHipScheduler::Task t = d.queue.front ();
    d.queue.pop_front ();
    // a reference that failed (or was skipped): this picture is skipped, as picture->error does in the reference
    bool skip = false;
    for (Ref * r:t.refs)
      skip |= r->failed;
    // references that live elsewhere: complete by now (runnable); bring their frames over -- once each
    // (refs = {n, n} is legal)
    struct Move {
      Ref *ref;
      void *frame;
      hipEvent_t ready;
      void *moved;
      hipEvent_t done;
    };
    std::vector < Move > to_move;
    for (Ref * r:t.refs) {
      bool listed = false;
      for (auto & m:to_move)
        listed |= m.ref == r;
      if (!skip && !listed && r->device != index && !r->copies.count (index))
        to_move.push_back ({ r, r->frame, r->ready, nullptr, nullptr });
    }
    d.current = &t;
    lock.unlock ();
    int rc = skip ? SCHRO_HIP_ESKIPPED : 0;
    for (size_t k = 0; k < to_move.size () && !rc; k++) {
      Move & m = to_move[k];
      if (!m.frame)
        continue;               // nothing published: the caller moves it (foreign_ref of submit)
      if (s->virtual_devices) {
        m.moved = m.frame;
      } else {
        // the copy waits for the owner's `ready` on this device's copy queue; this thread does not (measured, see above)
        if (hipEventCreateWithFlags (&m.done, hipEventDisableTiming) != hipSuccess)
          m.done = nullptr;
        m.moved = m.done ? schro::frame_copy_to_async (d.ctx, (SchroHipFrame *) m.frame, m.ready, m.done) : nullptr;
        if (!m.moved)
          rc = SCHRO_HIP_EDEVICE;
      }
    }
    if (!to_move.empty ()) {
      lock.lock ();
      for (auto & m:to_move)
        if (m.moved) {
          m.ref->copies[index] = m.moved;
          if (m.done)
            m.ref->copied.push_back (m.done);
          s->moves++;
        } else if (m.done) {
          (void) hipEventDestroy (m.done);
        }
      lock.unlock ();
    }
    if (!rc) {
      if (d.ctx)
        (void) schro_hip_context_select_queue (d.ctx, 0);        // (the selection of the picture before does not leak into this one)
      rc = t.func (d.ctx, index, t.priv);
    }
    // the function only ENQUEUES on the context's queues: a reference is complete -- usable by its
    // dependents, here through the in-order queues, elsewhere through `ready` -- when an event stands
    // behind that work.  Nothing is drained.
    hipEvent_t ready = nullptr;
    int in_flight = 0;
    if (d.ctx && t.self) {
      if (hipEventCreateWithFlags (&ready, hipEventDisableTiming) != hipSuccess)
        ready = nullptr;
      const int rs = ready ? schro::context_join_queues (d.ctx, ready) : SCHRO_HIP_EDEVICE;
      if (!rc)
        rc = rs;
      // how many reference pictures of this device are in flight now (their events not yet reached): events
      // of the statistic's own, recorded right behind `ready` -- `ready` itself belongs to the record and is
      // destroyed with it, possibly before the next reference picture of this device looks here
      while (!d.in_flight.empty () && hipEventQuery (d.in_flight.front ()) != hipErrorNotReady) {
        (void) hipEventDestroy (d.in_flight.front ());
        d.in_flight.pop_front ();
      }
      hipEvent_t stat = nullptr;
      if (ready && !rs && hipEventCreateWithFlags (&stat, hipEventDisableTiming) == hipSuccess) {
        if (hipEventRecord (stat, d.ctx->streams[0]) == hipSuccess)
          d.in_flight.push_back (stat);
        else
          (void) hipEventDestroy (stat);
      }
      in_flight = (int) d.in_flight.size ();
    }
    lock.lock ();
    d.current = nullptr;
    if (rc && rc != SCHRO_HIP_ESKIPPED && !s->first_error)
      s->first_error = rc;
    if (rc == SCHRO_HIP_ESKIPPED)
      s->skipped++;
    if (in_flight > s->refs_in_flight_max)
      s->refs_in_flight_max = in_flight;
    if (t.self) {
      t.self->ready = ready;
      t.self->failed = rc != 0;
      t.self->complete = true;
      release_if_unused (s, t.self);
    }
    for (Ref * r:t.refs) {
      r->users--;
      release_if_unused (s, r);
    }
    d.finished++;
    s->work.notify_all ();      // a picture waiting for this one as a foreign reference; garbage elsewhere
    s->done.notify_all ();
  }
}

SchroHipScheduler *
scheduler_new (const int *devices, int n_devices, bool virt)
{
  int avail = 0;
  if (!virt && hipGetDeviceCount (&avail) != hipSuccess)
    avail = 0;
  if (n_devices <= 0 && !devices)
    n_devices = virt ? 1 : avail;
  bool ok = n_devices > 0;
  for (int k = 0; ok && !virt && k < n_devices; k++)
    ok = (devices ? devices[k] : k) >= 0 && (devices ? devices[k] : k) < avail;
  if (!ok) {
    schro::set_error (SCHRO_HIP_EDEVICE, "scheduler_new: %d device(s) asked for, %d visible", n_devices, avail);
    return nullptr;
  }
  SchroHipScheduler *s = new SchroHipScheduler ();
  s->virtual_devices = virt;
  s->devs.resize ((size_t) n_devices);
  for (int k = 0; k < n_devices; k++) {
    s->devs[k].device = devices ? devices[k] : k;
    if (!virt) {
      // (not bound to this thread: the caller may own a context of its own; the workers bind theirs)
      s->devs[k].ctx = schro::context_new_unbound (s->devs[k].device);
      if (!s->devs[k].ctx) {
        for (int j = 0; j < k; j++)
          schro_hip_context_free (s->devs[j].ctx);
        delete s;
        return nullptr;
      }
    }
  }
  for (int k = 0; k < n_devices; k++)
    s->devs[k].thread = std::thread (worker, s, k);
  return s;
}

}                               // namespace

extern "C" {

SchroHipScheduler *
schro_hip_scheduler_new (int n_devices)
{
  return scheduler_new (nullptr, n_devices, false);
}

SchroHipScheduler *
schro_hip_scheduler_new_on (const int *devices, int n_devices)
{
  if (!devices || n_devices <= 0) {
    schro::set_error (SCHRO_HIP_EINVAL, "scheduler_new_on: no device list");
    return nullptr;
  }
  return scheduler_new (devices, n_devices, false);
}

SchroHipScheduler *
schro_hip_scheduler_new_virtual (int n_devices)
{
  return scheduler_new (nullptr, n_devices, true);
}

void
schro_hip_scheduler_free (SchroHipScheduler * s)
{
  if (!s)
    return;
  {
    std::unique_lock < std::mutex > lock (s->mutex);
    s->done.wait (lock,[&] {
          for (auto & d:s->devs)
            if (d.finished != d.submitted)
              return false;
          return true;
        });
    // what was never retired goes now
    for (auto & o:s->owner) {
      o.second->retired = true;
      o.second->users = 0;
      o.second->complete = true;
      release_if_unused (s, o.second);
    }
    s->owner.clear ();
    s->quit = true;
    s->work.notify_all ();
  }
  for (auto & d:s->devs)
    d.thread.join ();
  // (the caller's own domain binding, if it has one, is the caller's again afterwards)
  SchroHipContext *mine = schro_hip_thread_bound ();
  for (auto & d:s->devs) {
    if (d.ctx) {
      (void) hipSetDevice (d.device);
      schro_hip_thread_bind (d.ctx);
      empty_garbage (d.ctx, d.garbage, d.pending);
      poll_pending (d.pending, true);
      for (hipEvent_t ev:d.in_flight)
        (void) hipEventDestroy (ev);
      d.in_flight.clear ();
      schro_hip_context_free (d.ctx);
    }
  }
  schro_hip_thread_bind (mine);
  delete s;
}

int
schro_hip_scheduler_n_devices (SchroHipScheduler * s)
{
  return s ? (int) s->devs.size () : 0;
}

SchroHipContext *
schro_hip_scheduler_context (SchroHipScheduler * s, int index)
{
  return s && index >= 0 && index < (int) s->devs.size ()? s->devs[index].ctx : nullptr;
}

int
schro_hip_scheduler_submit (SchroHipScheduler * s, int picture_number, const int *refs, int n_refs, int is_ref,
    SchroHipPictureFunc func, void *priv, int *foreign_ref)
{
  if (!s || !func || n_refs < 0 || n_refs > 2 || (n_refs && !refs))
    return schro::set_error (SCHRO_HIP_EINVAL, "scheduler_submit: bad arguments");
  std::unique_lock < std::mutex > lock (s->mutex);
  int dev = -1, foreign = -1;
  SchroHipScheduler::Task t;
  for (int k = 0; k < n_refs; k++) {
    auto it = s->owner.find (refs[k]);
    if (it == s->owner.end ())
      return schro::set_error (SCHRO_HIP_EINVAL, "scheduler_submit: picture %d predicts from %d, which was "
          "never submitted as a reference (or is retired)", picture_number, refs[k]);
    if (dev < 0)
      dev = it->second->device;
    else if (it->second->device != dev)
      foreign = refs[k];        // a prediction across two chains
    t.refs.push_back (it->second);
  }
  if (dev < 0) {
    // a new chain: the device with the least work outstanding; ties to the device that holds the fewest live references
    // (their dependents are still to come: with fast workers two anchors in a row would both see idle devices), then to
    // the lowest index
    std::vector < long >live (s->devs.size (), 0);
    for (const auto & o : s->owner)
      if (!o.second->retired)
        live[(size_t) o.second->device]++;
    long best = -1, best_live = -1;
    for (int k = 0; k < (int) s->devs.size (); k++) {
      const long load = s->devs[k].submitted - s->devs[k].finished;
      if (dev < 0 || load < best || (load == best && live[(size_t) k] < best_live)) {
        dev = k;
        best = load;
        best_live = live[(size_t) k];
      }
    }
  }
  for (Ref * r:t.refs)
    r->users++;
  t.number = picture_number;
  t.self = nullptr;
  if (is_ref) {
    auto old = s->owner.find (picture_number);
    if (old != s->owner.end ()) {       // the number is reused: the old record is retired
      old->second->retired = true;
      release_if_unused (s, old->second);
    }
    Ref *r = new Ref ();
    r->number = picture_number;
    r->device = dev;
    s->owner[picture_number] = r;
    t.self = r;
  }
  t.func = func;
  t.priv = priv;
  s->devs[dev].queue.push_back (t);
  s->devs[dev].submitted++;
  if (foreign_ref)
    *foreign_ref = foreign;
  s->work.notify_all ();
  return dev;
}

int
schro_hip_scheduler_retire (SchroHipScheduler * s, int picture_number)
{
  if (!s)
    return SCHRO_HIP_EINVAL;
  std::unique_lock < std::mutex > lock (s->mutex);
  auto it = s->owner.find (picture_number);
  if (it == s->owner.end ())
    return 0;
  Ref *r = it->second;
  s->owner.erase (it);
  r->retired = true;
  release_if_unused (s, r);
  s->work.notify_all ();        // (frames to release on their devices' threads)
  return 0;
}

int
schro_hip_scheduler_publish_reference (SchroHipScheduler * s, int device_index, void *frame)
{
  if (!s || device_index < 0 || device_index >= (int) s->devs.size () || !frame)
    return schro::set_error (SCHRO_HIP_EINVAL, "scheduler_publish_reference: bad arguments");
  std::unique_lock < std::mutex > lock (s->mutex);
  const SchroHipScheduler::Task * t = s->devs[device_index].current;
  if (!t || !t->self)
    return schro::set_error (SCHRO_HIP_EINVAL, "scheduler_publish_reference: no reference picture is running on device %d",
        device_index);
  if (t->self->frame)
    return schro::set_error (SCHRO_HIP_EINVAL, "scheduler_publish_reference: picture %d has published its frame", t->number);
  t->self->frame = s->virtual_devices ? frame : (void *) schro_hip_frame_ref ((SchroHipFrame *) frame);
  return 0;
}

void *
schro_hip_scheduler_reference_frame (SchroHipScheduler * s, int device_index, int picture_number)
{
  if (!s || device_index < 0 || device_index >= (int) s->devs.size ())
    return nullptr;
  std::unique_lock < std::mutex > lock (s->mutex);
  const SchroHipScheduler::Task * t = s->devs[device_index].current;
  if (!t)
    return nullptr;
  for (Ref * r:t->refs) {
    if (r->number != picture_number)
      continue;
    if (r->device == device_index)
      return r->frame;
    auto c = r->copies.find (device_index);
    return c == r->copies.end ()? nullptr : c->second;
  }
  return nullptr;
}

long
schro_hip_scheduler_moves (SchroHipScheduler * s)
{
  if (!s)
    return 0;
  std::unique_lock < std::mutex > lock (s->mutex);
  return s->moves;
}

long
schro_hip_scheduler_skipped (SchroHipScheduler * s)
{
  if (!s)
    return 0;
  std::unique_lock < std::mutex > lock (s->mutex);
  return s->skipped;
}

int
schro_hip_scheduler_refs_in_flight_max (SchroHipScheduler * s)
{
  if (!s)
    return 0;
  std::unique_lock < std::mutex > lock (s->mutex);
  return s->refs_in_flight_max;
}

int
schro_hip_scheduler_wait (SchroHipScheduler * s)
{
  if (!s)
    return SCHRO_HIP_EINVAL;
  std::unique_lock < std::mutex > lock (s->mutex);
  s->done.wait (lock,[&] {
        for (auto & d:s->devs)
          if (d.finished != d.submitted)
            return false;
        return true;
      });
  const int r = s->first_error;
  s->first_error = 0;
  return r;
}

}                               // extern "C"
