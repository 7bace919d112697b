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
//     picture, every VC-2 low-delay picture) starts a chain on the least loaded device;
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
// Lifetime (r03).  The reference decoder retires a reference at PARSE time
// (schro_decoder_reference_retire, schrodecoder.c:1302), i.e. possibly before pictures that
// predict from it -- already submitted -- have run, and even before the reference itself has.  So
// a reference's state is not looked up by number when a dependent runs: submit resolves the
// numbers to state records once, a record counts the submitted pictures that still need it, and
// retire only removes the number from the lookup table; the record (and the frames it holds) goes
// when it is retired, complete and no longer needed.  Frames are released on the thread of the
// device whose context owns them (a context is not thread-safe).
#include "schro_hip_internal.h"

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

extern "C" SchroHipFrame *schro_hip_frame_copy_to (SchroHipContext * dst_ctx, SchroHipFrame * src);
extern "C" void schro_hip_thread_bind (SchroHipContext * ctx);

struct SchroHipScheduler {
  struct Ref {
    int number;
    int device;                 // index of the owner
    bool complete = false;      // the owner's function has returned AND its device work has finished
    bool retired = false;       // no longer in the lookup table
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
    std::vector < SchroHipFrame * >garbage;      // frames of this device's context to release on its thread
  };
  std::vector < Device > devs;
  std::mutex mutex;
  std::condition_variable work, done;
  std::map < int, Ref * >owner; // picture number -> record (references not yet retired)
  bool quit = false;
  bool virtual_devices = false;
  int first_error = 0;
  long moves = 0;
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
    if (r->frame)
      s->devs[r->device].garbage.push_back ((SchroHipFrame *) r->frame);
    for (auto & c:r->copies)
      s->devs[c.first].garbage.push_back ((SchroHipFrame *) c.second);
  }
  delete r;
}

void
empty_garbage (std::vector < SchroHipFrame * >&g)
{
  for (SchroHipFrame * f:g)
    schro_hip_frame_unref (f);
  g.clear ();
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
    s->work.wait (lock,[&] {
          return s->quit || runnable (s, d, index) || !d.garbage.empty ();
        });
    if (!d.garbage.empty ()) {
      std::vector < SchroHipFrame * >g;
      g.swap (d.garbage);
      lock.unlock ();
      empty_garbage (g);
      lock.lock ();
      continue;
    }
    if (!runnable (s, d, index)) {
      if (s->quit)
        return;
      continue;
    }
    SchroHipScheduler::Task t = d.queue.front ();
    d.queue.pop_front ();
    // references that live elsewhere: complete by now (runnable); bring their frames over
    std::vector < std::pair < Ref *, void *> >to_move;
    for (Ref * r:t.refs)
      if (r->device != index && !r->copies.count (index))
        to_move.push_back ({ r, r->frame });
    d.current = &t;
    lock.unlock ();
    int rc = 0;
    std::vector < void *>moved (to_move.size (), nullptr);
    for (size_t k = 0; k < to_move.size () && !rc; k++) {
      if (!to_move[k].second)
        continue;               // nothing published: the caller moves it (foreign_ref of submit)
      if (s->virtual_devices) {
        moved[k] = to_move[k].second;
      } else {
        moved[k] = schro_hip_frame_copy_to (d.ctx, (SchroHipFrame *) to_move[k].second);
        if (!moved[k])
          rc = SCHRO_HIP_EDEVICE;
      }
    }
    if (!to_move.empty ()) {
      lock.lock ();
      for (size_t k = 0; k < to_move.size (); k++)
        if (moved[k]) {
          to_move[k].first->copies[index] = moved[k];
          s->moves++;
        }
      lock.unlock ();
    }
    if (!rc)
      rc = t.func (d.ctx, index, t.priv);
    // the function only ENQUEUES on the context's queues: a reference counts as complete -- readable
    // from another device -- when that work has finished
    if (d.ctx && t.self) {
      const int rs = schro_hip_synchronize (d.ctx);
      if (!rc)
        rc = rs;
    }
    lock.lock ();
    d.current = nullptr;
    if (rc && !s->first_error)
      s->first_error = rc;
    if (t.self) {
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
      s->devs[k].ctx = schro_hip_context_new (s->devs[k].device);
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
  for (auto & d:s->devs) {
    if (d.ctx) {
      (void) hipSetDevice (d.device);
      schro_hip_thread_bind (d.ctx);
      empty_garbage (d.garbage);
      schro_hip_context_free (d.ctx);
    }
  }
  schro_hip_thread_bind (nullptr);
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
    // a new chain: the device with the least work outstanding, ties to the lowest index
    long best = -1;
    for (int k = 0; k < (int) s->devs.size (); k++) {
      const long load = s->devs[k].submitted - s->devs[k].finished;
      if (dev < 0 || load < best) {
        dev = k;
        best = load;
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
