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
//     a picture whose references ended up on two devices (a prediction across chains) waits for
//     the foreign reference's device and is reported to the caller, who moves the frame
//     (one peer copy of a u8 picture).
// No data-path collective, no RCCL: pictures shard.
#include "schro_hip_internal.h"

#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

struct SchroHipScheduler {
  struct Task {
    int number;
    int foreign_ref;            // picture number on another device to wait for, or -1
    SchroHipPictureFunc func;
    void *priv;
  };
  struct Device {
    SchroHipContext *ctx = nullptr;
    int device = -1;
    std::thread thread;
    std::deque < Task > queue;
    long submitted = 0, finished = 0;
  };
  std::vector < Device > devs;
  std::mutex mutex;
  std::condition_variable work, done;
  std::map < int, int >owner;           // picture number -> device index (pictures that are references)
  std::map < int, bool > complete;      // reference pictures whose function has returned
  bool quit = false;
  bool virtual_devices = false;
  int first_error = 0;
};

namespace {

void
worker (SchroHipScheduler * s, int index)
{
  SchroHipScheduler::Device & d = s->devs[index];
  if (d.ctx)
    (void) hipSetDevice (d.device);     // the exec domain of this thread
  std::unique_lock < std::mutex > lock (s->mutex);
  for (;;) {
    s->work.wait (lock,[&] {
          if (s->quit)
            return true;
          if (d.queue.empty ())
            return false;
          const int f = d.queue.front ().foreign_ref;
          return f < 0 || s->complete.count (f) != 0;
        });
    if (d.queue.empty ()) {
      if (s->quit)
        return;
      continue;
    }
    SchroHipScheduler::Task t = d.queue.front ();
    if (t.foreign_ref >= 0 && !s->complete.count (t.foreign_ref)) {
      if (s->quit)
        return;
      continue;
    }
    d.queue.pop_front ();
    lock.unlock ();
    const int r = t.func (d.ctx, index, t.priv);
    lock.lock ();
    if (r && !s->first_error)
      s->first_error = r;
    if (s->owner.count (t.number))
      s->complete[t.number] = true;
    d.finished++;
    s->work.notify_all ();      // a picture waiting for this one as a foreign reference
    s->done.notify_all ();
  }
}

SchroHipScheduler *
scheduler_new (int n_devices, bool virt)
{
  int avail = 0;
  if (!virt && hipGetDeviceCount (&avail) != hipSuccess)
    avail = 0;
  if (n_devices <= 0)
    n_devices = virt ? 1 : avail;
  if (n_devices <= 0 || (!virt && n_devices > avail)) {
    schro::set_error (SCHRO_HIP_EDEVICE, "scheduler_new: %d device(s) asked for, %d visible", n_devices, avail);
    return nullptr;
  }
  SchroHipScheduler *s = new SchroHipScheduler ();
  s->virtual_devices = virt;
  s->devs.resize ((size_t) n_devices);
  for (int k = 0; k < n_devices; k++) {
    s->devs[k].device = k;
    if (!virt) {
      s->devs[k].ctx = schro_hip_context_new (k);
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
  return scheduler_new (n_devices, false);
}

SchroHipScheduler *
schro_hip_scheduler_new_virtual (int n_devices)
{
  return scheduler_new (n_devices, true);
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
    s->quit = true;
    s->work.notify_all ();
  }
  for (auto & d:s->devs)
    d.thread.join ();
  for (auto & d:s->devs)
    if (d.ctx)
      schro_hip_context_free (d.ctx);
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
  for (int k = 0; k < n_refs; k++) {
    auto it = s->owner.find (refs[k]);
    if (it == s->owner.end ())
      return schro::set_error (SCHRO_HIP_EINVAL, "scheduler_submit: picture %d predicts from %d, which was "
          "never submitted as a reference", picture_number, refs[k]);
    if (dev < 0)
      dev = it->second;
    else if (it->second != dev)
      foreign = refs[k];        // a prediction across two chains
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
  if (is_ref) {
    s->owner[picture_number] = dev;
    s->complete.erase (picture_number);
  }
  s->devs[dev].queue.push_back (SchroHipScheduler::Task { picture_number, foreign, func, priv });
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
  s->owner.erase (picture_number);
  s->complete.erase (picture_number);
  return 0;
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
