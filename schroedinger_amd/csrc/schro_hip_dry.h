// schro_hip_dry.h -- the DEVICE-FREE build of the host code (-DSCHRO_HIP_DRY): test infrastructure for the sanitizers.
//
// SURVEY 5 lists race detection / sanitizers among the reference's auxiliary subsystems; GPU-side sanitizers are not
// available on this pool, and most of the library's host code -- job tables, tile orders and records, weight tables,
// wavelet and dequantisation geometry, the frame layer, the scheduler -- is index arithmetic that only ran behind a
// device.  In this build every HIP runtime entry point the host code uses is a host stand-in: "device" memory is heap
// memory (so AddressSanitizer sees every table write and every copy's bounds), copies are memcpy, queues and events
// complete at once, kernel launches are dropped (SCHRO_LAUNCH in schro_hip_internal.h).  Results are NOT computed -- the
// planes keep whatever they held --; what runs, and is checked by ASAN / UBSAN / TSAN, is everything in front of and
// around the launches, with the real arguments.  make -C schroedinger_amd/csrc dry_asan dry_tsan builds
// libschro_hip_dry_asan.so / _dry_tsan.so; tests/test_sanitizers.py drives them (CPU suite).  Never shipped, never
// loaded by the product or by bench.py.
#pragma once

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>

namespace schro_dry {

struct Registry {
  std::mutex m;
  std::set < const void *>device;       // allocations made through hipMalloc: what hipPointerGetAttributes calls device memory
};
inline Registry & registry ()
{
  static Registry r;
  return r;
}

inline hipError_t Malloc (void **p, size_t n)
{
  *p = malloc (n ? n : 1);
  if (!*p)
    return hipErrorOutOfMemory;
  std::lock_guard < std::mutex > lock (registry ().m);
  registry ().device.insert (*p);
  return hipSuccess;
}

template < typename T > inline hipError_t Malloc (T ** p, size_t n)
{
  return Malloc ((void **) p, n);
}

inline hipError_t Free (void *p)
{
  if (p) {
    std::lock_guard < std::mutex > lock (registry ().m);
    registry ().device.erase (p);
  }
  free (p);
  return hipSuccess;
}

inline hipError_t HostMalloc (void **p, size_t n, unsigned)
{
  *p = malloc (n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}

inline hipError_t HostFree (void *p)
{
  free (p);
  return hipSuccess;
}

inline hipError_t Copy2D (void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height)
{
  for (size_t y = 0; y < height; y++)
    memcpy ((char *) dst + y * dpitch, (const char *) src + y * spitch, width);
  return hipSuccess;
}

inline hipError_t PointerAttributes (hipPointerAttribute_t * a, const void *p)
{
  // an address inside one of "the device's" allocations is device memory; anything else is an ordinary host pointer,
  // which the real call answers with an error
  std::lock_guard < std::mutex > lock (registry ().m);
  auto it = registry ().device.upper_bound (p);
  if (it == registry ().device.begin ())
    return hipErrorInvalidValue;
  --it;
  // (sizes are not kept: the nearest allocation below stands for it -- good enough for "vectors already on the device")
  if ((const char *) p - (const char *) *it > (1 << 30))
    return hipErrorInvalidValue;
  memset (a, 0, sizeof (*a));
  a->type = hipMemoryTypeDevice;
  return hipSuccess;
}

inline hipError_t DeviceProperties (hipDeviceProp_t * prop, int)
{
  memset (prop, 0, sizeof (*prop));
  prop->multiProcessorCount = 256;
  strcpy (prop->name, "dry run (no device)");
  return hipSuccess;
}

inline hipError_t One (void **h)
{
  *h = malloc (1);              // (an event / a queue: a handle that can be told from NULL and freed)
  return hipSuccess;
}

inline hipError_t Drop (void *h)
{
  free (h);
  return hipSuccess;
}

}                               // namespace schro_dry

#define hipMalloc(p, n) schro_dry::Malloc (p, n)
#define hipFree(p) schro_dry::Free ((void *) (p))
#define hipHostMalloc(p, n, f) schro_dry::HostMalloc ((void **) (p), n, f)
#define hipHostFree(p) schro_dry::HostFree ((void *) (p))
#define hipSetDevice(d) ((void) (d), hipSuccess)
#define hipGetDeviceCount(n) (*(n) = 8, hipSuccess)
#undef hipGetDeviceProperties
#define hipGetDeviceProperties(p, d) schro_dry::DeviceProperties (p, d)
#define hipDeviceSynchronize() hipSuccess
#define hipGetLastError() hipSuccess
#define hipStreamCreateWithFlags(s, f) schro_dry::One ((void **) (s))
#define hipExtStreamCreateWithCUMask(s, w, m) schro_dry::One ((void **) (s))
#define hipStreamDestroy(s) schro_dry::Drop ((void *) (s))
#define hipStreamSynchronize(s) ((void) (s), hipSuccess)
#define hipStreamWaitEvent(s, e, f) ((void) (s), (void) (e), hipSuccess)
#define hipEventCreate(e) schro_dry::One ((void **) (e))
#define hipEventCreateWithFlags(e, f) schro_dry::One ((void **) (e))
#define hipEventDestroy(e) schro_dry::Drop ((void *) (e))
#define hipEventRecord(e, s) ((void) (e), (void) (s), hipSuccess)
#define hipEventSynchronize(e) ((void) (e), hipSuccess)
#define hipEventQuery(e) ((void) (e), hipSuccess)
#define hipEventElapsedTime(ms, a, b) (*(ms) = 0.0f, hipSuccess)
#define hipMemcpy(d, s, n, k) (memcpy (d, s, n), hipSuccess)
#define hipMemcpyAsync(d, s, n, k, q) (memcpy (d, s, n), hipSuccess)
#define hipMemcpyPeerAsync(d, dd, s, sd, n, q) (memcpy (d, s, n), hipSuccess)
#define hipMemcpy2DAsync(d, dp, s, sp, w, h, k, q) schro_dry::Copy2D (d, dp, s, sp, w, h)
#define hipMemset(d, v, n) (memset (d, v, n), hipSuccess)
#define hipMemsetAsync(d, v, n, q) (memset (d, v, n), hipSuccess)
#define hipPointerGetAttributes(a, p) schro_dry::PointerAttributes (a, p)
