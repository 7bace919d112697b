// iiwt_haar.hip -- one level of the 2-D inverse Haar wavelet (Dirac filters 3 and 4) on s32
// coefficients: the low-delay 10-bit configurations (BASELINE config 5: 7680x4320 4:2:2, s32,
// Haar without shift).
//
// What it computes: schro_iiwt_haar0 / _haar1 (schroedinger/schrowaveletorc.c:1697-1764) on 32-bit
// samples -- vertical lifting over the level view (even rows -= avgs (odd rows, 0); odd rows +=
// even rows: orc_haar_synth_s32), then per row the same two steps between the left and the right
// half and the interleave (orc_haar_synth_int_s32 / orc_haar_synth_rrshift1_int_s32, which halves
// every output with avgs (x, 0) for filter 4).  Every lifting tap of the Haar pair sits at offset
// 0, so a 2x2 block of output samples depends on exactly one sample of each sub-band: no halo, no
// neighbours, no LDS -- a lane loads 16 bytes of each sub-band row (4 columns), computes 4 x (2 x 2)
// outputs and stores two 32-byte row pieces.  The general LDS kernel (iiwt.hip) ran this level at
// 4.2 TB/s; this form is bound by HBM alone (8 B read + 8 B written per sub-band sample position,
// i.e. 4 B + 4 B per output sample).
//
// avgs (x, 0) = (x + 1) >> 1 computed without overflow = (x >> 1) + (x & 1); adds and subtracts wrap
// at 32 bits as the Orc programs' addl / subl do.

#include "schro_hip_internal.h"

namespace schro {
namespace {

constexpr int kHaarThreads = 256;
constexpr int kHaarCols = 64 * 4;       // sub-band columns per workgroup (one wave wide)
constexpr int kHaarRows = kHaarThreads / 64;    // sub-band rows (= output row pairs) per workgroup

__device__ __forceinline__ uint32_t
avgs0 (uint32_t x)
{
  return (uint32_t) (((int32_t) x >> 1) + (int32_t) (x & 1u));
}

// (a, b) <- inverse Haar pair: a -= avgs (b, 0); b += a
__device__ __forceinline__ void
haar_pair (uint32_t & a, uint32_t & b)
{
  a -= avgs0 (b);
  b += a;
}

template < int SHIFT >
__global__ __launch_bounds__ (kHaarThreads)
void iiwt_haar_s32_kernel (const IwtJob * __restrict__ jobs, int njobs)
{
  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  const IwtJob job = jobs[find_job (jobs, njobs, bid)];
  const int t = bid - job.tile_base;
  const int ty = t / job.tiles_x, tx = t - ty * job.tiles_x;
  const int nc = job.w / 2, nr = job.h / 2;
  const int col = (tx * 64 + (int) (threadIdx.x & 63)) * 4, row = ty * kHaarRows + (int) (threadIdx.x >> 6);
  if (col >= nc || row >= nr)
    return;
  u32x4 sb[4];
#pragma unroll
  for (int s = 0; s < 4; s++)
    sb[s] = gload < u32x4 > ((const char *) job.sb[s] + (size_t) row * job.sb_stride[s] + (size_t) col * 4);
  uint32_t even[8], odd[8];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    uint32_t ll = sb[0][c], hl = sb[1][c], lh = sb[2][c], hh = sb[3][c];
    haar_pair (ll, lh);         // vertical, left half: rows 2j / 2j + 1
    haar_pair (hl, hh);         // vertical, right half
    haar_pair (ll, hl);         // horizontal, even row
    haar_pair (lh, hh);         // horizontal, odd row
    if constexpr (SHIFT) {
      ll = avgs0 (ll);
      hl = avgs0 (hl);
      lh = avgs0 (lh);
      hh = avgs0 (hh);
    }
    even[2 * c] = ll;
    even[2 * c + 1] = hl;
    odd[2 * c] = lh;
    odd[2 * c + 1] = hh;
  }
  char *d0 = (char *) job.dst + (size_t) (2 * row) * job.dst_stride + (size_t) (2 * col) * 4;
  char *d1 = d0 + job.dst_stride;
  gstore < u32x4 > (d0, (u32x4) { even[0], even[1], even[2], even[3] });
  gstore < u32x4 > (d0 + 16, (u32x4) { even[4], even[5], even[6], even[7] });
  gstore < u32x4 > (d1, (u32x4) { odd[0], odd[1], odd[2], odd[3] });
  gstore < u32x4 > (d1 + 16, (u32x4) { odd[4], odd[5], odd[6], odd[7] });
}

}                               // namespace

// which levels this form takes: s32, Haar, every sub-band row and the destination 16-byte aligned,
// whole groups of four columns
bool
iiwt_haar_supported (int filter, int bpp)
{
  return bpp == 4 && (filter == 3 || filter == 4);
}

bool
iiwt_haar_job_ok (const IwtJob & j)
{
  uintptr_t bits = (uintptr_t) j.dst | (uintptr_t) j.dst_stride;
  for (int s = 0; s < 4; s++)
    bits |= (uintptr_t) j.sb[s] | (uintptr_t) j.sb_stride[s];
  return (bits & 15) == 0 && (j.w / 2) % 4 == 0 && j.w >= 8 && j.h >= 2;
}

void
iiwt_haar_geometry (int *cols, int *rows)
{
  *cols = kHaarCols;
  *rows = kHaarRows;
}

int
launch_iiwt_haar (hipStream_t stream, const IwtJob * d_jobs, int njobs, int total_tiles, int filter)
{
  if (filter == 3)
    hipLaunchKernelGGL ((iiwt_haar_s32_kernel < 0 >), dim3 (total_tiles), dim3 (kHaarThreads), 0, stream, d_jobs, njobs);
  else
    hipLaunchKernelGGL ((iiwt_haar_s32_kernel < 1 >), dim3 (total_tiles), dim3 (kHaarThreads), 0, stream, d_jobs, njobs);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "iiwt (Haar s32) launch: %s", hipGetErrorString (e));
  return 0;
}

}                               // namespace schro
