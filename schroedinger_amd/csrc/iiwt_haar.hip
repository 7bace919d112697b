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

// ---- r03: all three levels of a depth-3 Haar transform in one pass ------------------------------
// Without halos the whole synthesis is local: an 8x8 block of output samples depends on one LL2
// coefficient, one of each level-2 detail band, 2x2 of each level-1 band and 4x4 of each finest band --
// 64 coefficients, all in the 8 frame rows of the block (the in-place layout interleaves the levels'
// ROWS: level l's view is {w >> l, h >> l, stride << l}; its columns are split low | high).  A lane
// owns one block: 22 loads (4 dwords, 6 x 8 bytes, 12 x 16 bytes; adjacent lanes = adjacent blocks, so
// every load instruction reads contiguous runs), three levels of haar_pair in registers, 16 stores of
// 16 bytes.  The coefficient frame is read once and the pixels written once: 8 B per sample instead
// of the per-level launches' 10.5 (levels 1 and 2 went through the intermediate LL planes).
constexpr int kHaar3Rows = kHaarThreads / 64;    // rows of blocks per workgroup (64 blocks wide)

template < int SHIFT >
__device__ __forceinline__ void
haar_quad (uint32_t ll, uint32_t hl, uint32_t lh, uint32_t hh, uint32_t * o00, uint32_t * o01, uint32_t * o10, uint32_t * o11)
{
  haar_pair (ll, lh);
  haar_pair (hl, hh);
  haar_pair (ll, hl);
  haar_pair (lh, hh);
  if constexpr (SHIFT) {
    ll = avgs0 (ll);
    hl = avgs0 (hl);
    lh = avgs0 (lh);
    hh = avgs0 (hh);
  }
  *o00 = ll;
  *o01 = hl;
  *o10 = lh;
  *o11 = hh;
}

// (plain accesses: streaming loads and stores measured no different here, r03)
#define H3_LOAD(V, p) gload < V > (p)
#define H3_STORE(p, ...) gstore < u32x4 > ((p), (__VA_ARGS__))
template < int SHIFT >
__global__ __launch_bounds__ (kHaarThreads)
void iiwt_haar3_s32_kernel (const IwtJob * __restrict__ jobs, int njobs)
{
  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  const IwtJob job = jobs[find_job (jobs, njobs, bid)];
  const int t = bid - job.tile_base;
  const int ty = t / job.tiles_x, tx = t - ty * job.tiles_x;
  const int w = job.w, nbx = w / 8, nby = job.h / 8;
  const int bx = tx * 64 + (int) (threadIdx.x & 63), by = ty * kHaar3Rows + (int) (threadIdx.x >> 6);
  if (bx >= nbx || by >= nby)
    return;
  const char *base = (const char *) job.sb[0] + (size_t) (8 * by) * job.sb_stride[0];
  const size_t S = (size_t) job.sb_stride[0];
  // level 2 (the coarsest): one coefficient per band; LL2 / HL2 in frame row 8 by, LH2 / HH2 in row 8 by + 4
  const uint32_t ll2 = H3_LOAD (uint32_t, base + (size_t) bx * 4), hl2 = H3_LOAD (uint32_t, base + (size_t) (w / 8 + bx) * 4);
  const uint32_t lh2 = H3_LOAD (uint32_t, base + 4 * S + (size_t) bx * 4), hh2 = H3_LOAD (uint32_t, base + 4 * S + (size_t) (w / 8 + bx) * 4);
  // level 1: 2 x 2 per band; HL1 in rows 8 by + 4 i, LH1 / HH1 in rows 8 by + 4 i + 2
  u32x2 hl1[2], lh1[2], hh1[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    hl1[i] = H3_LOAD (u32x2, base + (size_t) (4 * i) * S + (size_t) (w / 4 + 2 * bx) * 4);
    lh1[i] = H3_LOAD (u32x2, base + (size_t) (4 * i + 2) * S + (size_t) (2 * bx) * 4);
    hh1[i] = H3_LOAD (u32x2, base + (size_t) (4 * i + 2) * S + (size_t) (w / 4 + 2 * bx) * 4);
  }
  // level 0: 4 x 4 per band; HL0 in rows 8 by + 2 i, LH0 / HH0 in rows 8 by + 2 i + 1
  u32x4 hl0[4], lh0[4], hh0[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    hl0[i] = H3_LOAD (u32x4, base + (size_t) (2 * i) * S + (size_t) (w / 2 + 4 * bx) * 4);
    lh0[i] = H3_LOAD (u32x4, base + (size_t) (2 * i + 1) * S + (size_t) (4 * bx) * 4);
    hh0[i] = H3_LOAD (u32x4, base + (size_t) (2 * i + 1) * S + (size_t) (w / 2 + 4 * bx) * 4);
  }
  uint32_t l1[2][2], l0[4][4], px[8][8];
  haar_quad < SHIFT > (ll2, hl2, lh2, hh2, &l1[0][0], &l1[0][1], &l1[1][0], &l1[1][1]);
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
      haar_quad < SHIFT > (l1[i][j], hl1[i][j], lh1[i][j], hh1[i][j], &l0[2 * i][2 * j], &l0[2 * i][2 * j + 1],
          &l0[2 * i + 1][2 * j], &l0[2 * i + 1][2 * j + 1]);
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
      haar_quad < SHIFT > (l0[i][j], hl0[i][j], lh0[i][j], hh0[i][j], &px[2 * i][2 * j], &px[2 * i][2 * j + 1],
          &px[2 * i + 1][2 * j], &px[2 * i + 1][2 * j + 1]);
  char *d = (char *) job.dst + (size_t) (8 * by) * job.dst_stride + (size_t) (8 * bx) * 4;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    H3_STORE (d + (size_t) r * job.dst_stride, (u32x4) { px[r][0], px[r][1], px[r][2], px[r][3] });
    H3_STORE (d + (size_t) r * job.dst_stride + 16, (u32x4) { px[r][4], px[r][5], px[r][6], px[r][7] });
  }
}

}                               // namespace

// the three-level form: a depth-3 s32 Haar transform of a plane whose rows and band origins are 16-byte
// aligned (job.sb[0] / sb_stride[0]: the coefficient plane; w, h: the plane's size)
bool
iiwt_haar3_job_ok (const void *src, int src_stride, const void *dst, int dst_stride, int w, int h)
{
  return ((((uintptr_t) src | (uintptr_t) src_stride | (uintptr_t) dst | (uintptr_t) dst_stride) & 15) == 0) && w % 32 == 0
      && h % 8 == 0 && w >= 32 && h >= 8;
}

void
iiwt_haar3_geometry (int *blocks_x, int *blocks_y)
{
  *blocks_x = 64;
  *blocks_y = kHaar3Rows;
}

int
launch_iiwt_haar3 (hipStream_t stream, const IwtJob * d_jobs, int njobs, int total_tiles, int filter)
{
  if (filter == 3)
    SCHRO_LAUNCH ((iiwt_haar3_s32_kernel < 0 >), dim3 (total_tiles), dim3 (kHaarThreads), 0, stream, d_jobs, njobs);
  else
    SCHRO_LAUNCH ((iiwt_haar3_s32_kernel < 1 >), dim3 (total_tiles), dim3 (kHaarThreads), 0, stream, d_jobs, njobs);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "iiwt (Haar s32, three levels) launch: %s", hipGetErrorString (e));
  return 0;
}

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
    SCHRO_LAUNCH ((iiwt_haar_s32_kernel < 0 >), dim3 (total_tiles), dim3 (kHaarThreads), 0, stream, d_jobs, njobs);
  else
    SCHRO_LAUNCH ((iiwt_haar_s32_kernel < 1 >), dim3 (total_tiles), dim3 (kHaarThreads), 0, stream, d_jobs, njobs);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "iiwt (Haar s32) launch: %s", hipGetErrorString (e));
  return 0;
}

}                               // namespace schro
