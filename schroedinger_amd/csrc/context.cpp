// context.cpp -- the C ABI of libschro_hip.so (include/schro_hip.h), first part: errors, the context and its
// memory domain (SchroMemoryDomain-shaped), queues / marks / copies, job-table caches, per-launch profiling.
// (r04: api.cpp split by layer -- context.cpp, plane.cpp (r05: plane_iiwt / _frameops / _lowdelay / _obmc.cpp): the batched plane-level launches, frame.cpp: the
// SchroFrame-shaped stage boundary.)  Host logic only; the kernels are in the .hip files.

#include "schro_hip_internal.h"
#include <mutex>
#include <set>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

namespace schro {

static thread_local char g_err[512] = "";
static int g_abort_on_error = 0;

int
set_error (int code, const char *fmt, ...)
{
  va_list ap;
  va_start (ap, fmt);
  vsnprintf (g_err, sizeof (g_err), fmt, ap);
  va_end (ap);
  if (g_abort_on_error) {
    // the reference's convention at this boundary: SCHRO_ASSERT -> abort
    fprintf (stderr, "schro_hip: %s\n", g_err);
    abort ();
  }
  return code;
}

int
set_status (int code, const char *fmt, ...)
{
  va_list ap;
  va_start (ap, fmt);
  vsnprintf (g_err, sizeof (g_err), fmt, ap);
  va_end (ap);
  return code;
}

int
push_args (SchroHipContext * ctx, const void *host, size_t bytes, void **dev)
{
  if (bytes == 0 || bytes > SchroHipContext::kArgSlotBytes)
    return set_error (SCHRO_HIP_EINVAL, "job table of %zu bytes exceeds a table slot", bytes);
  uint64_t h = 1469598103934665603ull;          // FNV-1a, 8 bytes at a time
  {
    const unsigned char *b = (const unsigned char *) host;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
      uint64_t w;
      memcpy (&w, b + i, 8);
      h = (h ^ w) * 1099511628211ull;
    }
    for (; i < bytes; i++)
      h = (h ^ b[i]) * 1099511628211ull;
  }
  ctx->arg_clock++;
  constexpr int per_queue = SchroHipContext::kArgSlots / SchroHipContext::kQueues;
  const int k0 = ctx->cur * per_queue;
  int victim = k0;
  for (int k = k0; k < k0 + per_queue; k++) {
    SchroHipContext::ArgSlot & sl = ctx->arg_slots[k];
    if (sl.bytes == bytes && sl.hash == h
        && memcmp (ctx->h_args + (size_t) k * SchroHipContext::kArgSlotBytes, host, bytes) == 0) {
      sl.last_use = ctx->arg_clock;
      *dev = ctx->d_args + (size_t) k * SchroHipContext::kArgSlotBytes;
      return 0;
    }
    if (sl.last_use < ctx->arg_slots[victim].last_use)
      victim = k;
  }
  SchroHipContext::ArgSlot & sl = ctx->arg_slots[victim];
  char *hm = ctx->h_args + (size_t) victim * SchroHipContext::kArgSlotBytes;
  char *dm = ctx->d_args + (size_t) victim * SchroHipContext::kArgSlotBytes;
  if (sl.copy_pending) {        // the mirror may still be the source of an enqueued copy
    SCHRO_HIP_CHECK (hipEventSynchronize (sl.copied));
    sl.copy_pending = false;
  }
  memcpy (hm, host, bytes);
  sl.bytes = 0;
  SCHRO_HIP_CHECK (hipMemcpyAsync (dm, hm, bytes, hipMemcpyHostToDevice, ctx->stream));
  SCHRO_HIP_CHECK (hipEventRecord (sl.copied, ctx->stream));
  sl.copy_pending = true;
  sl.hash = h;
  sl.bytes = bytes;
  sl.last_use = ctx->arg_clock;
  *dev = dm;
  return 0;
}

// ... in two steps, for a caller that builds the table in place: *host is the pinned mirror to fill with
// `bytes` bytes, big_table_commit sends it
int
big_table_begin (SchroHipContext * ctx, size_t bytes, void **host, void **dev)
{
  const int q = ctx->cur;
  ctx->big_turn[q] = (ctx->big_turn[q] + 1) % SchroHipContext::kBigTables;
  SchroHipContext::BigTable & b = ctx->big_q[q][ctx->big_turn[q]];
  if (b.pending) {              // the mirror may still be the source of its last copy
    SCHRO_HIP_CHECK (hipEventSynchronize (b.copied));
    b.pending = false;
  }
  if (bytes > b.cap) {
    if (b.cap) {
      SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));     // launches that still read the old table
      (void) hipHostFree (b.h);
      (void) hipFree (b.d);
      b.cap = 0;
    }
    const size_t cap = (std::max (bytes + bytes / 2, (size_t) 256 << 10) + 15) & ~(size_t) 15;
    SCHRO_HIP_CHECK (hipHostMalloc ((void **) &b.h, cap, hipHostMallocDefault));
    SCHRO_HIP_CHECK (hipMalloc ((void **) &b.d, cap));
    if (!b.copied)
      SCHRO_HIP_CHECK (hipEventCreateWithFlags (&b.copied, hipEventDisableTiming));
    b.cap = cap;
  }
  *host = b.h;
  *dev = b.d;
  return 0;
}

int
big_table_commit (SchroHipContext * ctx, size_t bytes)
{
  SchroHipContext::BigTable & b = ctx->big_q[ctx->cur][ctx->big_turn[ctx->cur]];
  // (a kernel, not hipMemcpyAsync: the runtime's copy path blocks the calling thread when the queue waits for another
  // queue's event)
  const int r = launch_table_copy (ctx->stream, b.d, b.h, bytes);
  if (r)
    return r;
  SCHRO_HIP_CHECK (hipEventRecord (b.copied, ctx->stream));
  b.pending = true;
  return 0;
}

int
push_big_table (SchroHipContext * ctx, const void *host, size_t bytes, void **dev)
{
  void *mirror;
  int r = big_table_begin (ctx, bytes, &mirror, dev);
  if (r)
    return r;
  memcpy (mirror, host, bytes);
  return big_table_commit (ctx, bytes);
}

int
ensure_scratch (SchroHipContext * ctx, size_t bytes)
{
  void *&scratch = ctx->scratch_q[ctx->cur];
  size_t & size = ctx->scratch_size_q[ctx->cur];
  if (bytes <= size)
    return 0;
  if (scratch) {
    SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
    SCHRO_HIP_CHECK (hipFree (scratch));
    scratch = nullptr;
    size = 0;
  }
  SCHRO_HIP_CHECK (hipMalloc (&scratch, bytes));
  size = bytes;
  return 0;
}

// Reported ONCE, by the first call that looks (the launch's epoch names the batch it belongs to); the word is
// cleared so that later launches of the context are judged on their own.  (The kernel's counters need no
// reset: they are tagged with the launch's epoch.)
int
dc_gave_up (SchroHipContext * ctx)
{
  if (ctx->dc_gave_up && *(volatile uint32_t *) ctx->dc_gave_up) {
    const uint32_t epoch = *(volatile uint32_t *) ctx->dc_gave_up;
    *(volatile uint32_t *) ctx->dc_gave_up = 0;
    return set_error (SCHRO_HIP_EDEVICE, "DC prediction launch %u: a strip gave up waiting for the strip above it (its band is incomplete)",
        epoch);
  }
  if (ctx->dc_gave_up && ((volatile uint32_t *) ctx->dc_gave_up)[1]) {
    const uint32_t epoch = ((volatile uint32_t *) ctx->dc_gave_up)[1];
    ((volatile uint32_t *) ctx->dc_gave_up)[1] = 0;
    return set_error (SCHRO_HIP_EDEVICE, "inverse wavelet launch %u: a tile gave up waiting for the level above it (its picture is incomplete)",
        epoch);
  }
  return 0;
}

// r06 -- predictions that did not fit 8 bits (prediction_only OBMC batches, numbered).  Not an error of the
// stream -- the reference decodes such a picture (16-bit wrap, schromotion8.c:542-657) -- but an answer the caller
// routes on: the picture of that batch is wrong and takes the residual order.  Never aborts.  Ring word k belongs to ONE
// batch at a time (ovf_epoch[k]) and is looked at only when that batch's event has fired, so a flag is never
// attributed to another batch; what is found goes into two lists and stays there until it has been named by a
// synchronising call (once) and fetched by schro_hip_obmc_overflowed.
static void
pred_overflow_collect (SchroHipContext * ctx, int k, bool wait)
{
  if (!ctx->dc_gave_up || !ctx->ovf_epoch[k] || !ctx->ovf_ev[k])
    return;
  if (wait)
    (void) hipEventSynchronize (ctx->ovf_ev[k]);
  else if (hipEventQuery (ctx->ovf_ev[k]) != hipSuccess) {
    (void) hipGetLastError ();  // (hipErrorNotReady is not an error of ours)
    return;
  }
  volatile uint32_t *ring = (volatile uint32_t *) ctx->dc_gave_up + 4;
  if (ring[k]) {
    ring[k] = 0;
    ctx->ovf_unannounced.push_back (ctx->ovf_epoch[k]);
    ctx->ovf_unfetched.push_back (ctx->ovf_epoch[k]);
  }
  ctx->ovf_epoch[k] = 0;
}

int
pred_overflow_claim (SchroHipContext * ctx, uint32_t epoch)
{
  const int k = (int) (epoch % SchroHipContext::kOvfRing);
  // (the host is kOvfRing prediction_only batches ahead of the device only if nobody synchronises: then it waits here)
  pred_overflow_collect (ctx, k, true);
  if (!ctx->ovf_ev[k])
    SCHRO_HIP_CHECK (hipEventCreateWithFlags (&ctx->ovf_ev[k], hipEventDisableTiming));
  ctx->ovf_epoch[k] = epoch;
  return 0;
}

int
pred_overflow_poll (SchroHipContext * ctx)
{
  for (int k = 0; k < SchroHipContext::kOvfRing; k++)
    pred_overflow_collect (ctx, k, false);
  if (ctx->ovf_unannounced.empty ())
    return 0;
  char list[160];
  size_t n = 0;
  for (size_t i = 0; i < ctx->ovf_unannounced.size () && n + 16 < sizeof (list); i++)
    n += (size_t) snprintf (list + n, sizeof (list) - n, i ? ", %u" : "%u", ctx->ovf_unannounced[i]);
  ctx->ovf_unannounced.clear ();
  return set_status (SCHRO_HIP_ENEEDS_RESIDUAL, "prediction_only OBMC batch(es) %s met a DC value outside [-128, 127]: the prediction "
      "does not fit 8 bits and the combined picture differs from the reference's; such pictures take the residual order "
      "(schro_hip_obmc_overflowed returns the batches' numbers)", list);
}

extern "C" int
schro_hip_obmc_overflowed (SchroHipContext * ctx, unsigned int *epochs, int max)
{
  SCHRO_HIP_REQUIRE (ctx && (epochs || max == 0) && max >= 0, "obmc_overflowed: bad arguments");
  for (int k = 0; k < SchroHipContext::kOvfRing; k++)
    pred_overflow_collect (ctx, k, false);
  const int n = (int) std::min ((size_t) max, ctx->ovf_unfetched.size ());
  for (int i = 0; i < n; i++) {
    epochs[i] = ctx->ovf_unfetched[(size_t) i];
    // (what the caller has been handed need not interrupt a later synchronising call)
    auto it = std::find (ctx->ovf_unannounced.begin (), ctx->ovf_unannounced.end (), epochs[i]);
    if (it != ctx->ovf_unannounced.end ())
      ctx->ovf_unannounced.erase (it);
  }
  ctx->ovf_unfetched.erase (ctx->ovf_unfetched.begin (), ctx->ovf_unfetched.begin () + n);
  return n;
}

int
dc_edge_for (SchroHipContext * ctx, int njobs, int max_rows, int max_w, unsigned long long **edge, int *edge_pitch,
    uint32_t * epoch)
{
  if (!ctx->dc_gave_up) {
    SCHRO_HIP_CHECK (hipHostMalloc ((void **) &ctx->dc_gave_up, 64, hipHostMallocDefault));
    memset (ctx->dc_gave_up, 0, 64);
  }
  {
    const int r = dc_gave_up (ctx);
    if (r)
      return r;
  }
  void *&buf = ctx->dc_edge_q[ctx->cur];
  size_t & size = ctx->dc_edge_size_q[ctx->cur];
  const int strips = (max_rows + 63) / 64;
  // (8 words in front: the launch's ticket and finish counters)
  const size_t bytes = ((size_t) njobs * strips * max_w + 8) * sizeof (unsigned long long);
  if (bytes > size) {
    if (buf) {
      SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
      SCHRO_HIP_CHECK (hipFree (buf));
      buf = nullptr;
      size = 0;
    }
    SCHRO_HIP_CHECK (hipMalloc (&buf, bytes));
    size = bytes;
    // tag 0 = "never written": epochs start at 1
    SCHRO_HIP_CHECK (hipMemsetAsync (buf, 0, bytes, ctx->stream));
  }
  if (++ctx->dc_epoch == 0) {   // the counter wrapped: old tags must not look new
    for (int q = 0; q < SchroHipContext::kQueues; q++)
      if (ctx->dc_edge_q[q]) {
        SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->streams[q]));
        SCHRO_HIP_CHECK (hipMemset (ctx->dc_edge_q[q], 0, ctx->dc_edge_size_q[q]));
      }
    ctx->dc_epoch = 1;
  }
  *edge = (unsigned long long *) buf;
  *edge_pitch = max_w;
  *epoch = ctx->dc_epoch;
  return 0;
}

static thread_local SchroHipContext *t_prof_ctx = nullptr;
static thread_local int t_prof_cls = 0;

ProfileScope::ProfileScope (SchroHipContext * c, int cls)
{
  t_prof_ctx = c->profile ? c : nullptr;
  t_prof_cls = cls;
}

ProfileScope::~ProfileScope ()
{
  t_prof_ctx = nullptr;
}

bool
profile_launch_events (hipEvent_t * start, hipEvent_t * stop)
{
  SchroHipContext *c = t_prof_ctx;
  if (!c)
    return false;
  if (c->ev_used == c->ev_pool.size ()) {
    if (c->ev_pool.size () >= 16384)
      return false;             // pool exhausted: stop sampling, keep running
    SchroHipContext::EvPair p;
    if (hipEventCreate (&p.a) != hipSuccess)
      return false;
    if (hipEventCreate (&p.b) != hipSuccess) {
      (void) hipEventDestroy (p.a);
      return false;
    }
    c->ev_pool.push_back (p);
  }
  const size_t idx = c->ev_used++;
  c->ev_pool[idx].cls = t_prof_cls;
  *start = c->ev_pool[idx].a;
  *stop = c->ev_pool[idx].b;
  return true;
}

}                               // namespace schro

using namespace schro;

static_assert (sizeof (ObmcJob) * kMaxJobs <= SchroHipContext::kArgSlotBytes, "a launch group of kMaxJobs OBMC jobs fits a table slot");

extern "C" {

// ---- context / domain ---------------------------------------------------------

int
schro_hip_device_count (void)
{
  int n = 0;
  if (hipGetDeviceCount (&n) != hipSuccess)
    return 0;
  return n;
}

const char *
schro_hip_last_error (void)
{
  return g_err;
}

void
schro_hip_set_abort_on_error (int enable)
{
  g_abort_on_error = enable;
}

// schro_cuda_init (schrocuda.c:13-31): look at the devices; nothing else to set up
void
schro_hip_init (void)
{
  int n = 0;
  if (hipGetDeviceCount (&n) != hipSuccess)
    n = 0;
  for (int i = 0; i < n; i++) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties (&prop, i) == hipSuccess && getenv ("SCHRO_HIP_DEBUG"))
      fprintf (stderr, "schro_hip: device %d: %s %s, %d CUs, %zu MB\n", i, prop.name, prop.gcnArchName,
          prop.multiProcessorCount, prop.totalGlobalMem >> 20);
  }
}

// ---- the SchroMemoryDomain-shaped handle -------------------------------------------------------
// alloc / free of the reference's table carry no domain argument (schrodomain.h:18-22), so they
// resolve the domain from the calling thread's current device, as the reference's CUDA table
// resolves its device from the CUDA runtime's current-device state.
// r03: "current" is per THREAD, set by schro_hip_thread_bind -- schro_hip_context_new binds the thread that
// creates a context, the scheduler's exec-domain threads bind theirs -- and a thread that never bound
// one gets an error instead of, silently, device 0's domain (the HIP runtime's per-thread default).
static thread_local SchroHipContext *t_bound_ctx = nullptr;
// r04: the contexts that exist.  A thread may still be bound to a context another thread has freed
// (schro_hip_context_free can only clear the CALLING thread's binding): the table's alloc then fails
// loudly instead of following a dangling pointer.
static std::mutex g_live_mutex;
static std::set < SchroHipContext * >g_live;

static SchroHipContext *
current_device_context ()
{
  SchroHipContext *ctx = t_bound_ctx;
  if (!ctx)
    return nullptr;
  std::lock_guard < std::mutex > lock (g_live_mutex);
  if (!g_live.count (ctx)) {
    t_bound_ctx = nullptr;      // freed by another thread
    return nullptr;
  }
  return ctx;
}

static void *
domain_vt_alloc (int size)
{
  SchroHipContext *ctx = current_device_context ();
  if (!ctx || size <= 0) {
    set_error (SCHRO_HIP_EINVAL, "domain alloc (%d): this thread is not an exec-domain thread of a HIP domain "
        "(schro_hip_thread_bind)", size);
    fprintf (stderr, "schro_hip: %s\n", schro_hip_last_error ());
    return nullptr;
  }
  (void) hipSetDevice (ctx->device);
  // plain allocation: the caller (schro_memory_domain_alloc, schrodomain.c:58-109) keeps its own
  // slot cache on top of this table
  void *p = nullptr;
  if (hipMalloc (&p, (size_t) size) != hipSuccess) {
    set_error (SCHRO_HIP_ENOMEM, "domain alloc: hipMalloc (%d) failed", size);
    return nullptr;
  }
  return p;
}

static void *
domain_vt_alloc_2d (int depth, int width, int height)
{
  // the reference's CUDA table returns a cudaArray for texture fetches here; this path has no
  // textures (coordinate clamp in the kernels), so a 2-D block is a linear one
  if (depth <= 0 || width <= 0 || height <= 0)
    return nullptr;
  return domain_vt_alloc (((depth + 7) / 8) * width * height);
}

static void
domain_vt_free (void *ptr, int size)
{
  (void) size;
  if (ptr)
    (void) hipFree (ptr);
}

void
schro_hip_thread_bind (SchroHipContext * ctx)
{
  t_bound_ctx = ctx;
  if (ctx)
    (void) hipSetDevice (ctx->device);
}

SchroHipContext *
schro_hip_thread_bound (void)
{
  return current_device_context ();
}

// ---- pinned host memory: what the DMA engines copy from / to at full rate and asynchronously ----
void *
schro_hip_host_alloc (size_t size)
{
  void *p = nullptr;
  if (size == 0 || hipHostMalloc (&p, size, hipHostMallocDefault) != hipSuccess) {
    set_error (SCHRO_HIP_ENOMEM, "host_alloc (%zu) failed", size);
    return nullptr;
  }
  return p;
}

void
schro_hip_host_free (void *ptr)
{
  if (ptr)
    (void) hipHostFree (ptr);
}

static void *
host_vt_alloc (int size)
{
  return size > 0 ? schro_hip_host_alloc ((size_t) size) : nullptr;
}

static void *
host_vt_alloc_2d (int depth, int width, int height)
{
  if (depth <= 0 || width <= 0 || height <= 0)
    return nullptr;
  return host_vt_alloc (((depth + 7) / 8) * width * height);
}

static void
host_vt_free (void *ptr, int size)
{
  (void) size;
  schro_hip_host_free (ptr);
}

// A SchroMemoryDomain whose blocks are pinned HOST memory (flags: SCHRO_MEMORY_DOMAIN_CPU): frames the
// reference allocates in it (schro_frame_new_and_alloc (domain, ...): the transform frames the
// arithmetic decoder writes, the output pictures) are ordinary host frames to every CPU stage and the
// source / destination of asynchronous copies for this library.
SchroHipMemoryDomain *
schro_memory_domain_new_hip_host (void)
{
  SchroHipMemoryDomain *d = (SchroHipMemoryDomain *) calloc (1, sizeof (SchroHipMemoryDomain));
  d->flags = 0x0001;            // SCHRO_MEMORY_DOMAIN_CPU, schrodomain.h:34
  d->alloc = host_vt_alloc;
  d->alloc_2d = host_vt_alloc_2d;
  d->free = host_vt_free;
  d->ctx = nullptr;
  return d;
}

}                               // extern "C"

namespace schro {
// A context that does not touch the calling thread's domain binding (the scheduler's contexts belong to its
// worker threads; the thread that creates the scheduler may own a context of its own).
SchroHipContext *
context_new_unbound (int device)
{
  if (hipSetDevice (device) != hipSuccess) {
    set_error (SCHRO_HIP_EDEVICE, "hipSetDevice(%d) failed", device);
    return nullptr;
  }
  SchroHipContext *ctx = new SchroHipContext ();
  ctx->device = device;
  ctx->domain = (SchroHipMemoryDomain *) calloc (1, sizeof (SchroHipMemoryDomain));
  ctx->domain->flags = SCHRO_MEMORY_DOMAIN_HIP;
  ctx->domain->alloc = domain_vt_alloc;
  ctx->domain->alloc_2d = domain_vt_alloc_2d;
  ctx->domain->free = domain_vt_free;
  ctx->domain->ctx = ctx;
  ctx->domain_bytes = 0;
  for (int q = 0; q < SchroHipContext::kQueues; q++) {
    ctx->scratch_q[q] = nullptr;
    ctx->dc_edge_q[q] = nullptr;
    ctx->dc_edge_size_q[q] = 0;
    memset (ctx->big_q[q], 0, sizeof (ctx->big_q[q]));
    ctx->big_turn[q] = 0;
    ctx->scratch_size_q[q] = 0;
    ctx->dc_epoch = 0;
    ctx->dc_gave_up = nullptr;
    ctx->pred_epoch = 0;
    ctx->frame_dq_plan = nullptr;
    ctx->dq_stage_q[q] = nullptr;
    ctx->dq_stage_size_q[q] = 0;
    ctx->pack_tmp_q[q] = nullptr;
    ctx->pack_tmp_size_q[q] = 0;
    memset (ctx->ovf_epoch, 0, sizeof (ctx->ovf_epoch));
    memset (ctx->ovf_ev, 0, sizeof (ctx->ovf_ev));
    ctx->streams[q] = nullptr;
    ctx->queue_ev[q] = nullptr;
  }
  {
    hipDeviceProp_t prop;
    ctx->cus = hipGetDeviceProperties (&prop, device) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  ctx->cur = 0;
  ctx->stage_complete = true;
  memset (ctx->chain_slots, 0, sizeof (ctx->chain_slots));
  memset (ctx->chain_ctrl, 0, sizeof (ctx->chain_ctrl));
  memset (ctx->chain_ctrl_words, 0, sizeof (ctx->chain_ctrl_words));
  memset (ctx->chain_ctrl_hash, 0, sizeof (ctx->chain_ctrl_hash));
  memset (ctx->chain_runs, 0, sizeof (ctx->chain_runs));
  memset (ctx->marks, 0, sizeof (ctx->marks));
  ctx->arg_clock = 0;
  memset (ctx->arg_slots, 0, sizeof (ctx->arg_slots));
  memset (ctx->order_slots, 0, sizeof (ctx->order_slots));
  ctx->profile = false;
  ctx->ev_used = 0;
  ctx->h_args = nullptr;
  ctx->d_args = nullptr;
  bool ok = true;
  for (int q = 0; ok && q < SchroHipContext::kQueues; q++)
    ok = hipStreamCreateWithFlags (&ctx->streams[q], hipStreamNonBlocking) == hipSuccess
        && hipEventCreateWithFlags (&ctx->queue_ev[q], hipEventDisableTiming) == hipSuccess;
  ctx->stream = ctx->streams[0];
  ok = ok && hipEventCreate (&ctx->ev_begin) == hipSuccess
      && hipEventCreate (&ctx->ev_end) == hipSuccess
      && hipHostMalloc ((void **) &ctx->h_args, SchroHipContext::kArgSlots * SchroHipContext::kArgSlotBytes,
          hipHostMallocDefault) == hipSuccess
      && hipMalloc ((void **) &ctx->d_args, SchroHipContext::kArgSlots * SchroHipContext::kArgSlotBytes) == hipSuccess;
  for (int k = 0; ok && k < SchroHipContext::kArgSlots; k++)
    ok = hipEventCreateWithFlags (&ctx->arg_slots[k].copied, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    set_error (SCHRO_HIP_EDEVICE, "context creation failed on device %d: %s", device,
        hipGetErrorString (hipGetLastError ()));
    free (ctx->domain);
    delete ctx;
    return nullptr;
  }
  {
    std::lock_guard < std::mutex > lock (g_live_mutex);
    g_live.insert (ctx);
  }
  return ctx;
}

// Queue 0 waits for everything enqueued so far on the context's other queues, then `ev` is recorded on it:
// one event behind all the device work of a picture, whichever queues its function used.  The second kernel
// queue then waits for `ev` too (r05, ADVICE r04: "pictures of the same device follow the reference in the
// in-order queues" was true of queue 0 only) -- kernels behind an unfired event cost the host nothing.  The
// copy queues do NOT wait: a copy enqueued behind an unfired event holds its caller, and what a
// later picture uploads is ordered against earlier readers by that picture's own marks.
// No host wait.
int
context_join_queues (SchroHipContext * ctx, hipEvent_t ev)
{
  (void) hipSetDevice (ctx->device);
  for (int q = 1; q < SchroHipContext::kQueues; q++) {
    SCHRO_HIP_CHECK (hipEventRecord (ctx->queue_ev[q], ctx->streams[q]));
    SCHRO_HIP_CHECK (hipStreamWaitEvent (ctx->streams[0], ctx->queue_ev[q], 0));
  }
  SCHRO_HIP_CHECK (hipEventRecord (ev, ctx->streams[0]));
  SCHRO_HIP_CHECK (hipStreamWaitEvent (ctx->streams[1], ev, 0));
  return 0;
}
}                               // namespace schro

extern "C" {

// The creating thread becomes the context's exec-domain thread -- unless it is one already (r03 rebound it
// unconditionally: a thread that owned device 0's context and created a second one found its argument-less
// alloc table serving the wrong device).
SchroHipContext *
schro_hip_context_new (int device)
{
  SchroHipContext *ctx = schro::context_new_unbound (device);
  if (ctx && !current_device_context ())
    t_bound_ctx = ctx;
  return ctx;
}

SchroHipMemoryDomain *
schro_memory_domain_new_hip (int device)
{
  SchroHipContext *ctx = schro_hip_context_new (device);
  return ctx ? ctx->domain : nullptr;
}

void
schro_memory_domain_free_hip (SchroHipMemoryDomain * domain)
{
  if (domain && (domain->flags & SCHRO_MEMORY_DOMAIN_HIP))
    schro_hip_context_free (domain->ctx);
}

SchroHipContext *
schro_hip_domain_context (SchroHipMemoryDomain * domain)
{
  return domain && (domain->flags & SCHRO_MEMORY_DOMAIN_HIP) ? domain->ctx : nullptr;
}

SchroHipMemoryDomain *
schro_hip_context_domain (SchroHipContext * ctx)
{
  return ctx ? ctx->domain : nullptr;
}

void
schro_hip_context_free (SchroHipContext * ctx)
{
  if (!ctx)
    return;
  (void) hipSetDevice (ctx->device);
  for (int q = 0; q < SchroHipContext::kQueues; q++)
    if (ctx->streams[q])
      (void) hipStreamSynchronize (ctx->streams[q]);
  if (ctx->frame_dq_plan) {
    schro_hip_dequant_plan_free (ctx->frame_dq_plan);
    ctx->frame_dq_plan = nullptr;
  }
  for (auto & s : ctx->slots)
    (void) hipFree (s.ptr);
  if (ctx->dc_gave_up)
    (void) hipHostFree (ctx->dc_gave_up);
  for (int k = 0; k < SchroHipContext::kOvfRing; k++)
    if (ctx->ovf_ev[k])
      (void) hipEventDestroy (ctx->ovf_ev[k]);
  for (int q = 0; q < SchroHipContext::kQueues; q++) {
    if (ctx->dq_stage_q[q])
      (void) hipFree (ctx->dq_stage_q[q]);
    if (ctx->pack_tmp_q[q])
      (void) hipFree (ctx->pack_tmp_q[q]);
    if (ctx->scratch_q[q])
      (void) hipFree (ctx->scratch_q[q]);
    if (ctx->dc_edge_q[q])
      (void) hipFree (ctx->dc_edge_q[q]);
    for (auto & b : ctx->big_q[q]) {
      if (b.cap) {
        (void) hipHostFree (b.h);
        (void) hipFree (b.d);
      }
      if (b.copied)
        (void) hipEventDestroy (b.copied);
    }
  }
  for (int k = 0; k < SchroHipContext::kOrderSlots; k++) {
    if (ctx->order_slots[k].d)
      (void) hipFree (ctx->order_slots[k].d);
    if (ctx->order_slots[k].h)
      (void) hipHostFree (ctx->order_slots[k].h);
    if (ctx->order_slots[k].copied)
      (void) hipEventDestroy (ctx->order_slots[k].copied);
  }
  for (int k = 0; k < SchroHipContext::kChainSlots; k++)
    if (ctx->chain_slots[k].d)
      (void) hipFree (ctx->chain_slots[k].d);
  for (int q = 0; q < SchroHipContext::kQueues; q++)
    if (ctx->chain_ctrl[q])
      (void) hipFree (ctx->chain_ctrl[q]);
  for (int k = 0; k < SchroHipContext::kArgSlots; k++)
    if (ctx->arg_slots[k].copied)
      (void) hipEventDestroy (ctx->arg_slots[k].copied);
  if (ctx->d_args)
    (void) hipFree (ctx->d_args);
  if (ctx->h_args)
    (void) hipHostFree (ctx->h_args);
  for (auto & p : ctx->ev_pool) {
    (void) hipEventDestroy (p.a);
    (void) hipEventDestroy (p.b);
  }
  (void) hipEventDestroy (ctx->ev_begin);
  (void) hipEventDestroy (ctx->ev_end);
  // (threads still bound to this context elsewhere find it gone: current_device_context)
  {
    std::lock_guard < std::mutex > lock (g_live_mutex);
    g_live.erase (ctx);
  }
  if (t_bound_ctx == ctx)
    t_bound_ctx = nullptr;
  free (ctx->domain);
  for (int m = 0; m < SchroHipContext::kMarks; m++)
    if (ctx->marks[m])
      (void) hipEventDestroy (ctx->marks[m]);
  for (int q = 0; q < SchroHipContext::kQueues; q++) {
    if (ctx->queue_ev[q])
      (void) hipEventDestroy (ctx->queue_ev[q]);
    if (ctx->streams[q])
      (void) hipStreamDestroy (ctx->streams[q]);
  }
  delete ctx;
}

// schro_memory_domain_alloc, schrodomain.c:58-103: reuse a free slot of
// exactly this size, else allocate a new one; nothing is returned to the
// device before the domain dies (schrodomain.c:105-137).
void *
schro_hip_domain_alloc (SchroHipContext * ctx, size_t size)
{
  if (!ctx || size == 0) {
    set_error (SCHRO_HIP_EINVAL, "domain_alloc: bad arguments");
    return nullptr;
  }
  for (auto & s : ctx->slots) {
    if (!s.in_use && s.size == size) {
      s.in_use = true;
      return s.ptr;
    }
  }
  void *p = nullptr;
  (void) hipSetDevice (ctx->device);
  hipError_t e = hipMalloc (&p, size);
  if (e != hipSuccess) {
    set_error (SCHRO_HIP_ENOMEM, "hipMalloc(%zu): %s", size, hipGetErrorString (e));
    return nullptr;
  }
  ctx->slots.push_back ({p, size, true});
  ctx->domain_bytes += size;
  return p;
}

int
schro_hip_domain_free (SchroHipContext * ctx, void *ptr)
{
  if (!ctx)
    return set_error (SCHRO_HIP_EINVAL, "domain_free: no context");
  for (auto & s : ctx->slots) {
    if (s.ptr == ptr && s.in_use) {
      s.in_use = false;
      return 0;
    }
  }
  return set_error (SCHRO_HIP_EINVAL, "domain_free: %p is not a live block of this domain", ptr);
}

size_t
schro_hip_domain_bytes (SchroHipContext * ctx)
{
  return ctx ? ctx->domain_bytes : 0;
}

int
schro_hip_upload_2d (SchroHipContext * ctx, void *dst, int dst_stride, const void *src,
    int src_stride, int row_bytes, int height)
{
  SCHRO_HIP_REQUIRE (ctx && dst && src && row_bytes > 0 && height > 0, "upload_2d: bad arguments");
  SCHRO_HIP_CHECK (hipMemcpy2DAsync (dst, dst_stride, src, src_stride, row_bytes, height,
          hipMemcpyHostToDevice, ctx->stream));
  SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
  return 0;
}

int
schro_hip_download_2d (SchroHipContext * ctx, void *dst, int dst_stride, const void *src,
    int src_stride, int row_bytes, int height)
{
  SCHRO_HIP_REQUIRE (ctx && dst && src && row_bytes > 0 && height > 0, "download_2d: bad arguments");
  SCHRO_HIP_CHECK (hipMemcpy2DAsync (dst, dst_stride, src, src_stride, row_bytes, height,
          hipMemcpyDeviceToHost, ctx->stream));
  SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
  return 0;
}

// The asynchronous forms: enqueued on the SELECTED queue (by convention SCHRO_HIP_QUEUE_H2D / _D2H), no
// wait.  With pinned host memory (schro_hip_host_alloc, schro_memory_domain_new_hip_host) the copy runs
// on a DMA engine beside the kernels of the other queues; order it with marks.  Rows that are
// contiguous on both sides go as ONE linear copy (the rectangle form is slower on the DMA engines).
}                               // extern "C"

namespace schro {
int
copy_2d_async (SchroHipContext * ctx, void *dst, int dst_stride, const void *src, int src_stride, int row_bytes,
    int height, hipMemcpyKind kind)
{
  (void) hipSetDevice (ctx->device);
  if (height == 1 || (dst_stride == row_bytes && src_stride == row_bytes))
    SCHRO_HIP_CHECK (hipMemcpyAsync (dst, src, (size_t) row_bytes * height, kind, ctx->stream));
  else
    SCHRO_HIP_CHECK (hipMemcpy2DAsync (dst, dst_stride, src, src_stride, row_bytes, height, kind, ctx->stream));
  return 0;
}
}                               // namespace schro

extern "C" {

int
schro_hip_upload_2d_async (SchroHipContext * ctx, void *dst, int dst_stride, const void *src,
    int src_stride, int row_bytes, int height)
{
  SCHRO_HIP_REQUIRE (ctx && dst && src && row_bytes > 0 && height > 0, "upload_2d_async: bad arguments");
  return copy_2d_async (ctx, dst, dst_stride, src, src_stride, row_bytes, height, hipMemcpyHostToDevice);
}

int
schro_hip_download_2d_async (SchroHipContext * ctx, void *dst, int dst_stride, const void *src,
    int src_stride, int row_bytes, int height)
{
  SCHRO_HIP_REQUIRE (ctx && dst && src && row_bytes > 0 && height > 0, "download_2d_async: bad arguments");
  return copy_2d_async (ctx, dst, dst_stride, src, src_stride, row_bytes, height, hipMemcpyDeviceToHost);
}

int
schro_hip_queue_synchronize (SchroHipContext * ctx, int queue)
{
  SCHRO_HIP_REQUIRE (ctx && queue >= 0 && queue < SchroHipContext::kQueues, "queue_synchronize: queue %d out of range", queue);
  SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->streams[queue]));
  const int r = dc_gave_up (ctx);
  return r ? r : pred_overflow_poll (ctx);
}

int
schro_hip_queue_set_cu_mask (SchroHipContext * ctx, int queue, const uint32_t * mask, int words)
{
  SCHRO_HIP_REQUIRE (ctx && queue >= 0 && queue < SchroHipContext::kQueues && mask && words > 0,
      "queue_set_cu_mask: bad arguments");
  (void) hipSetDevice (ctx->device);
  SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->streams[queue]));
  hipStream_t fresh;
  SCHRO_HIP_CHECK (hipExtStreamCreateWithCUMask (&fresh, (uint32_t) words, mask));
  (void) hipStreamDestroy (ctx->streams[queue]);
  ctx->streams[queue] = fresh;
  if (ctx->cur == queue)
    ctx->stream = fresh;
  return 0;
}

int
schro_hip_memset (SchroHipContext * ctx, void *dst, int value, size_t bytes)
{
  SCHRO_HIP_REQUIRE (ctx && dst, "memset: bad arguments");
  SCHRO_HIP_CHECK (hipMemsetAsync (dst, value, bytes, ctx->stream));
  return 0;
}

int
schro_hip_synchronize (SchroHipContext * ctx)
{
  SCHRO_HIP_REQUIRE (ctx, "synchronize: no context");
  for (int q = 0; q < SchroHipContext::kQueues; q++)
    SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->streams[q]));
  const int r = dc_gave_up (ctx);
  return r ? r : pred_overflow_poll (ctx);
}

// Queues.  The reference's scheduler runs the stages of different pictures on several worker
// threads at once (schroasync-pthread.c:320-390); on a GPU domain the same concurrency is two
// in-order queues: select one, enqueue a picture batch's stage calls, and order stages of
// different queues with schro_hip_queue_wait where one consumes what the other produced.
int
schro_hip_context_select_queue (SchroHipContext * ctx, int queue)
{
  SCHRO_HIP_REQUIRE (ctx && queue >= 0 && queue < SchroHipContext::kQueues, "select_queue: queue %d out of range", queue);
  ctx->cur = queue;
  ctx->stream = ctx->streams[queue];
  return 0;
}

int
schro_hip_context_queue (SchroHipContext * ctx)
{
  return ctx ? ctx->cur : -1;
}

// Marks: a finer dependency than "everything so far".  Picture batch k + 2's wavelet may
// overwrite batch k's residual frames once batch k's OBMC is done -- not batch k + 1's, which
// is what is last on that queue by then.
int
schro_hip_queue_mark (SchroHipContext * ctx, int mark)
{
  SCHRO_HIP_REQUIRE (ctx && mark >= 0 && mark < SchroHipContext::kMarks, "queue_mark: mark %d out of range", mark);
  (void) hipSetDevice (ctx->device);
  if (!ctx->marks[mark])
    SCHRO_HIP_CHECK (hipEventCreateWithFlags (&ctx->marks[mark], hipEventDisableTiming));
  SCHRO_HIP_CHECK (hipEventRecord (ctx->marks[mark], ctx->stream));
  return 0;
}

// the HOST waits for the latest recording of a mark (e.g. "picture k's download"): what a host that keeps
// several pictures in flight calls before it hands picture k on, instead of draining a whole queue
int
schro_hip_queue_mark_synchronize (SchroHipContext * ctx, int mark)
{
  SCHRO_HIP_REQUIRE (ctx && mark >= 0 && mark < SchroHipContext::kMarks, "queue_mark_synchronize: mark %d out of range", mark);
  if (!ctx->marks[mark])
    return 0;                   // never recorded: nothing to wait for
  SCHRO_HIP_CHECK (hipEventSynchronize (ctx->marks[mark]));
  return 0;
}

int
schro_hip_queue_wait_mark (SchroHipContext * ctx, int mark)
{
  SCHRO_HIP_REQUIRE (ctx && mark >= 0 && mark < SchroHipContext::kMarks, "queue_wait_mark: mark %d out of range", mark);
  if (!ctx->marks[mark])
    return 0;                   // never set: nothing to wait for
  (void) hipSetDevice (ctx->device);
  SCHRO_HIP_CHECK (hipStreamWaitEvent (ctx->stream, ctx->marks[mark], 0));
  return 0;
}

int
schro_hip_queue_wait (SchroHipContext * ctx, int waiter, int signaller)
{
  SCHRO_HIP_REQUIRE (ctx && waiter >= 0 && waiter < SchroHipContext::kQueues && signaller >= 0
      && signaller < SchroHipContext::kQueues, "queue_wait: bad arguments");
  if (waiter == signaller)
    return 0;
  (void) hipSetDevice (ctx->device);
  SCHRO_HIP_CHECK (hipEventRecord (ctx->queue_ev[signaller], ctx->streams[signaller]));
  SCHRO_HIP_CHECK (hipStreamWaitEvent (ctx->streams[waiter], ctx->queue_ev[signaller], 0));
  return 0;
}

void *
schro_hip_stream (SchroHipContext * ctx)
{
  return ctx ? (void *) ctx->stream : nullptr;
}

int
schro_hip_timer_begin (SchroHipContext * ctx)
{
  SCHRO_HIP_REQUIRE (ctx, "timer: no context");
  SCHRO_HIP_CHECK (hipEventRecord (ctx->ev_begin, ctx->stream));
  return 0;
}

float
schro_hip_timer_end (SchroHipContext * ctx)
{
  if (!ctx)
    return -1.f;
  float ms = -1.f;
  if (hipEventRecord (ctx->ev_end, ctx->stream) != hipSuccess
      || hipEventSynchronize (ctx->ev_end) != hipSuccess
      || hipEventElapsedTime (&ms, ctx->ev_begin, ctx->ev_end) != hipSuccess) {
    set_error (SCHRO_HIP_EDEVICE, "timer_end failed");
    return -1.f;
  }
  return ms;
}

int
schro_hip_profile_enable (SchroHipContext * ctx, int enable)
{
  SCHRO_HIP_REQUIRE (ctx, "profile: no context");
  ctx->profile = enable != 0;
  return 0;
}

int
schro_hip_profile_reset (SchroHipContext * ctx)
{
  SCHRO_HIP_REQUIRE (ctx, "profile: no context");
  SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
  ctx->ev_used = 0;
  return 0;
}

int
schro_hip_profile_read (SchroHipContext * ctx, int kernel_class, double *total_ms, int *launches)
{
  SCHRO_HIP_REQUIRE (ctx && total_ms && launches && kernel_class >= 0
      && kernel_class < SCHRO_HIP_KERNEL_CLASSES, "profile_read: bad arguments");
  SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
  double sum = 0;
  int n = 0;
  for (size_t i = 0; i < ctx->ev_used; i++) {
    if (ctx->ev_pool[i].cls != kernel_class)
      continue;
    float ms = 0;
    SCHRO_HIP_CHECK (hipEventElapsedTime (&ms, ctx->ev_pool[i].a, ctx->ev_pool[i].b));
    sum += ms;
    n++;
  }
  *total_ms = sum;
  *launches = n;
  return 0;
}

}                               // extern "C"
