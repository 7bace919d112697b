// frameops.hip -- the two frame-level kernels around the wavelet and OBMC:
//
//   convert:  dst_u8 = sat_u8 (s16|s32 + 128), cropped
//             schro_frame_convert -> convert_u8_s16 (schrovirtframe.c:1689-1720),
//             orc_offsetconvert_u8_s16 / _s32 (schroorc.orc:504-521)
//   upsample: the three half-pel planes of a reference component,
//             schro_upsampled_frame_upsample (schroframe.c:2000-2030):
//             v-half = 8-tap vertical, h-half = 8-tap horizontal,
//             hv-half = 8-tap horizontal of the v-half; taps {-1,3,-7,21,21,-7,3,-1},
//             clamp ((sum + 16) >> 5, 0, 255), border indices clamped, last
//             row / last column copied (schroframe.c:1552-1553, 1642-1644).
//
// Both are pure streaming kernels (HBM bound): convert reads 2|4 B and writes
// 1 B per sample; upsample reads 1 B and writes 4 B per sample into ONE
// interleaved 2w x 2h image (layout in include/schro_hip.h) so that the OBMC
// kernel's bilinear taps are neighbours in memory.

#include "schro_hip_internal.h"

namespace schro {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ int
clampi (int x, int lo, int hi)
{
  return min (max (x, lo), hi);
}

template < typename JOB >
__device__ __forceinline__ int
find_job (const JOB * jobs, int njobs, int bid)
{
  int j = 0;
  while (j + 1 < njobs && bid >= jobs[j + 1].tile_base)
    j++;
  return j;
}

// ---- convert ---------------------------------------------------------------

constexpr int kCvtTW = 512, kCvtTH = 4;

template < typename T >
__device__ __forceinline__ uint8_t
offsetconvert (T s)
{
  if constexpr (sizeof (T) == 2) {
    int v = (int16_t) (s + 128);                // addw wraps
    return (uint8_t) clampi (v, 0, 255);        // convsuswb
  } else {
    int t = (int) ((unsigned) s + 128u);        // addl
    t = clampi (t, -32768, 32767);              // convssslw
    return (uint8_t) clampi (t, 0, 255);
  }
}

template < typename T >
__global__ __launch_bounds__ (kThreads)
void convert_kernel (const ConvertJob * __restrict__ jobs, int njobs)
{
  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  const ConvertJob job = jobs[find_job (jobs, njobs, bid)];
  const int t = bid - job.tile_base;
  const int tx = t % job.tiles_x, ty = t / job.tiles_x;
  const int x = tx * kCvtTW + (threadIdx.x % 64) * 8;
  const int y = ty * kCvtTH + threadIdx.x / 64;
  if (y >= job.h || x >= job.w)
    return;
  const T *s = (const T *) ((const char *) job.src + (size_t) y * job.src_stride) + x;
  uint8_t *d = job.dst + (size_t) y * job.dst_stride + x;
  const bool vec = x + 8 <= job.w && (((uintptr_t) s & 15) == 0) && (((uintptr_t) d & 7) == 0);
  if (vec) {
    T v[8];
    if constexpr (sizeof (T) == 2) {
      *reinterpret_cast < uint4 * >(v) = *reinterpret_cast < const uint4 * >(s);
    } else {
      reinterpret_cast < uint4 * >(v)[0] = reinterpret_cast < const uint4 * >(s)[0];
      reinterpret_cast < uint4 * >(v)[1] = reinterpret_cast < const uint4 * >(s)[1];
    }
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      lo |= (uint32_t) offsetconvert < T > (v[e]) << (8 * e);
      hi |= (uint32_t) offsetconvert < T > (v[4 + e]) << (8 * e);
    }
    *reinterpret_cast < uint2 * >(d) = make_uint2 (lo, hi);
  } else {
    for (int e = 0; e < 8 && x + e < job.w; e++)
      d[e] = offsetconvert < T > (s[e]);
  }
}

// ---- upsample ----------------------------------------------------------------

constexpr int kUpTW = 64, kUpTH = 16;
constexpr int kUpLW = kUpTW + 8;        // 7 halo columns, padded

__device__ __forceinline__ int
mas8 (const int *s)
{
  // taps {-1, 3, -7, 21, 21, -7, 3, -1}, (x + 16) >> 5, clamp
  int x = 21 * (s[3] + s[4]) - 7 * (s[2] + s[5]) + 3 * (s[1] + s[6]) - (s[0] + s[7]);
  return clampi ((x + 16) >> 5, 0, 255);
}

__global__ __launch_bounds__ (kThreads)
void upsample_kernel (const UpsampleJob * __restrict__ jobs, int njobs)
{
  __shared__ uint8_t p0[kUpTH + 7][kUpLW];      // integer pels, rows y0-3 .. y0+TH+3
  __shared__ uint8_t p2[kUpTH][kUpLW];          // v-half, cols x0-3 .. x0+TW+3

  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  const UpsampleJob job = jobs[find_job (jobs, njobs, bid)];
  const int t = bid - job.tile_base;
  const int tx = t % job.tiles_x, ty = t / job.tiles_x;
  const int x0 = tx * kUpTW, y0 = ty * kUpTH;
  const int w = job.w, h = job.h;
  const int tid = threadIdx.x;

  // picture coordinates are clamped on the way in, which is the index clamp
  // of mas8_u8_edgeextend / the CLAMP (i + j - 3, 0, height - 1) row list
  for (int it = tid; it < (kUpTH + 7) * (kUpTW + 7); it += kThreads) {
    int lx = it % (kUpTW + 7), ly = it / (kUpTW + 7);
    int gx = clampi (x0 - 3 + lx, 0, w - 1);
    int gy = clampi (y0 - 3 + ly, 0, h - 1);
    p0[ly][lx] = job.src[(size_t) gy * job.src_stride + gx];
  }
  __syncthreads ();

  for (int it = tid; it < kUpTH * (kUpTW + 7); it += kThreads) {
    int lx = it % (kUpTW + 7), ly = it / (kUpTW + 7);
    int gy = y0 + ly;
    int s[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
      s[k] = p0[ly + k][lx];
    // last row of the v-half is a copy of the source row (schroframe.c:1642-1644)
    p2[ly][lx] = (gy >= h - 1) ? p0[ly + 3][lx] : (uint8_t) mas8 (s);
  }
  __syncthreads ();

  const int lx4 = (tid % 16) * 4;       // 4 pixels per thread
  const int ly = tid / 16;
  const int gy = y0 + ly;
  if (gy >= h)
    return;
  uint8_t row_e[8], row_o[8];           // HP rows 2*gy and 2*gy+1
  int valid = 0;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    int lx = lx4 + e;
    int gx = x0 + lx;
    if (gx >= w)
      break;
    valid = e + 1;
    int s0[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      s0[k] = p0[ly + 3][lx + k];
      s2[k] = p2[ly][lx + k];
    }
    int c0 = s0[3], c2 = s2[3];
    // last column: copy (mas8_u8_edgeextend d[n-1] = s[n-1]; for n <= 8 the
    // following schro_frame_mc_edgeextend_horiz overwrites it the same way)
    int c1 = (gx >= w - 1) ? c0 : mas8 (s0);
    int c3 = (gx >= w - 1) ? c2 : mas8 (s2);
    if (gy >= h - 1)
      c3 = c1;                  // last row of hv-half comes from the h-half (schroframe.c:2028)
    row_e[2 * e] = (uint8_t) c0;
    row_e[2 * e + 1] = (uint8_t) c1;
    row_o[2 * e] = (uint8_t) c2;
    row_o[2 * e + 1] = (uint8_t) c3;
  }
  if (!valid)
    return;
  uint8_t *de = job.dst + (size_t) (2 * gy) * job.dst_stride + 2 * (x0 + lx4);
  uint8_t *dod = de + job.dst_stride;
  if (valid == 4 && (((uintptr_t) de | (uintptr_t) dod) & 7) == 0) {
    *reinterpret_cast < uint2 * >(de) = *reinterpret_cast < const uint2 * >(row_e);
    *reinterpret_cast < uint2 * >(dod) = *reinterpret_cast < const uint2 * >(row_o);
  } else {
    for (int e = 0; e < 2 * valid; e++) {
      de[e] = row_e[e];
      dod[e] = row_o[e];
    }
  }
}

}                               // namespace

void
convert_tile_geometry (int *tw, int *th)
{
  *tw = kCvtTW;
  *th = kCvtTH;
}

int
launch_convert (hipStream_t stream, const ConvertJob * d_jobs, int njobs, int total_tiles, int bpp)
{
  if (bpp == 2)
    hipLaunchKernelGGL ((convert_kernel < int16_t >), dim3 (total_tiles), dim3 (kThreads), 0,
        stream, d_jobs, njobs);
  else
    hipLaunchKernelGGL ((convert_kernel < int32_t >), dim3 (total_tiles), dim3 (kThreads), 0,
        stream, d_jobs, njobs);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "convert launch: %s", hipGetErrorString (e));
  return 0;
}

void
upsample_tile_geometry (int *tw, int *th)
{
  *tw = kUpTW;
  *th = kUpTH;
}

int
launch_upsample (hipStream_t stream, const UpsampleJob * d_jobs, int njobs, int total_tiles)
{
  hipLaunchKernelGGL (upsample_kernel, dim3 (total_tiles), dim3 (kThreads), 0, stream, d_jobs,
      njobs);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "upsample launch: %s", hipGetErrorString (e));
  return 0;
}

}                               // namespace schro
