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


// ---- r05: the three levels AND the v210 copy-out in one pass (BASELINE config 5's whole pixel path) --------------
// The reference's chain for a 10-bit 4:2:2 intra picture (schrodecoder.c:1855-1886 x_wavelet_transform into the s32
// frame; x_combine :2011-2052 schro_frame_convert (output_picture, frame) = convert_s16_s32 (truncation) -> the crop /
// edge-extend identity -> pack_v210_s16, schrovirtframe.c:1438-1537, :943-991: clamp (x + 512, 0, 1023), 6 pixels in
// 16 bytes) writes 4 B per sample and reads them back: 2 x 265 MB per 8K picture around 88 MB of v210.  Here a
// workgroup owns a strip of 8 picture rows x 960 pixels of ALL THREE components: 120 lanes run the Y blocks, 60 the U
// and 60 the V blocks (8 x 8 samples each, the same code on another plane: a lane only differs in its base pointer and
// width), the 10-bit values meet in LDS (30 KB: u16 rows), and after one barrier every lane packs pairs of v210 groups
// -- 12 Y + 6 U + 6 V values in, 32 contiguous bytes out.  Per picture 265 MB read + 88 MB written instead of 265 +
// 265 + 265 + 88.
constexpr int kHpRegions = 20;                  // 48-pixel regions per workgroup: 6 Y + 3 U + 3 V blocks each
constexpr int kHpWidth = 48 * kHpRegions;       // 960 luma pixels
constexpr int kHpYBlocks = 6 * kHpRegions, kHpCBlocks = 3 * kHpRegions;

template < int SHIFT >
__global__ __launch_bounds__ (kHaarThreads)
void iiwt_haar3_v210_kernel (const HaarPackJob * __restrict__ jobs, int njobs)
{
  __shared__ __attribute__ ((aligned (16))) uint16_t s_y[8][kHpWidth];
  __shared__ __attribute__ ((aligned (16))) uint16_t s_c[2][8][kHpWidth / 2];
  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  // (the jobs' first tiles: a wave probes 64 at once, as find_job does for the other job types)
  int ji = 0;
  for (int k = 1; k < njobs; k++)
    ji = bid >= jobs[k].tile_base ? k : ji;
  const HaarPackJob job = jobs[ji];
  const int t = bid - job.tile_base;
  const int ty = t / job.tiles_x, tx = t - ty * job.tiles_x;
  const int tid = threadIdx.x;
  // which block of which component: lanes 0 .. 119 Y, 120 .. 179 U, 180 .. 239 V
  const int comp = tid < kHpYBlocks ? 0 : tid < kHpYBlocks + kHpCBlocks ? 1 : 2;
  const int lb = comp == 0 ? tid : comp == 1 ? tid - kHpYBlocks : tid - kHpYBlocks - kHpCBlocks;    // block within the strip piece
  const int w = comp ? job.w >> 1 : job.w;       // the component's transform width
  const int bx = tx * (comp ? kHpCBlocks : kHpYBlocks) + lb;
  const bool have = tid < kHpYBlocks + 2 * kHpCBlocks && bx < w / 8;
  if (have) {
    // (selects, not job.src[comp]: an array indexed by a per-lane value goes to scratch memory)
    const char *plane = (const char *) (comp == 0 ? job.src[0] : comp == 1 ? job.src[1] : job.src[2]);
    const size_t S = (size_t) (comp == 0 ? job.src_stride[0] : comp == 1 ? job.src_stride[1] : job.src_stride[2]);
    const char *base = plane + (size_t) (8 * ty) * S;
    const uint32_t ll2 = H3_LOAD (uint32_t, base + (size_t) bx * 4), hl2 = H3_LOAD (uint32_t, base + (size_t) (w / 8 + bx) * 4);
    const uint32_t lh2 = H3_LOAD (uint32_t, base + 4 * S + (size_t) bx * 4), hh2 = H3_LOAD (uint32_t, base + 4 * S + (size_t) (w / 8 + bx) * 4);
    u32x2 hl1[2], lh1[2], hh1[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      hl1[i] = H3_LOAD (u32x2, base + (size_t) (4 * i) * S + (size_t) (w / 4 + 2 * bx) * 4);
      lh1[i] = H3_LOAD (u32x2, base + (size_t) (4 * i + 2) * S + (size_t) (2 * bx) * 4);
      hh1[i] = H3_LOAD (u32x2, base + (size_t) (4 * i + 2) * S + (size_t) (w / 4 + 2 * bx) * 4);
    }
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
    // the sample pack_v210_s16 sees: the s32 value truncated to 16 bits (convert_s16_s32: convlw), + 512, clamped to 10 bits
    uint16_t *row0 = comp == 0 ? &s_y[0][8 * lb] : &s_c[comp - 1][0][8 * lb];
    const int pitch = comp == 0 ? kHpWidth : kHpWidth / 2;
#pragma unroll
    for (int r = 0; r < 8; r++) {
      uint32_t q[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int a = min (max ((int) (int16_t) px[r][2 * k] + 512, 0), 1023), b = min (max ((int) (int16_t) px[r][2 * k + 1] + 512, 0), 1023);
        q[k] = (uint32_t) a | ((uint32_t) b << 16);
      }
      *reinterpret_cast < u32x4 * >(row0 + (size_t) r * pitch) = (u32x4) { q[0], q[1], q[2], q[3] };
    }
  }
  __syncthreads ();
  // ---- pack: a task = 12 pixels of one row = two v210 groups = 32 bytes ----
  const int x0 = tx * kHpWidth, npx = min (kHpWidth, job.w - x0);       // (a multiple of 48: the host checks)
  const int pairs = npx / 12;
  for (int task = tid; task < 8 * pairs; task += kHaarThreads) {
    const int y = task / pairs, p = task - y * pairs;
    const u32x2 *yp = reinterpret_cast < const u32x2 * >(&s_y[y][12 * p]);
    const uint32_t *up = reinterpret_cast < const uint32_t * >(&s_c[0][y][6 * p]), *vp = reinterpret_cast < const uint32_t * >(&s_c[1][y][6 * p]);
    const u32x2 y01 = yp[0], y23 = yp[1], y45 = yp[2];
    const uint32_t yy[6] = { y01.x, y01.y, y23.x, y23.y, y45.x, y45.y };        // two pixels per word
    const uint32_t uu[3] = { up[0], up[1], up[2] }, vv[3] = { vp[0], vp[1], vp[2] };
    u32x4 o[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
      // group g: Y 6 g .. 6 g + 5 (words 3 g .. 3 g + 2), U / V 3 g .. 3 g + 2 (16-bit values 3 g, 3 g + 1, 3 g + 2)
      const uint32_t ya = yy[3 * g], yb = yy[3 * g + 1], yc = yy[3 * g + 2];
      const uint32_t y0 = ya & 0xffffu, y1 = ya >> 16, y2 = yb & 0xffffu, y3 = yb >> 16, y4 = yc & 0xffffu, y5 = yc >> 16;
      uint32_t cb0, cb1, cb2, cr0, cr1, cr2;
      if (g == 0) {
        cb0 = uu[0] & 0xffffu; cb1 = uu[0] >> 16; cb2 = uu[1] & 0xffffu;
        cr0 = vv[0] & 0xffffu; cr1 = vv[0] >> 16; cr2 = vv[1] & 0xffffu;
      } else {
        cb0 = uu[1] >> 16; cb1 = uu[2] & 0xffffu; cb2 = uu[2] >> 16;
        cr0 = vv[1] >> 16; cr1 = vv[2] & 0xffffu; cr2 = vv[2] >> 16;
      }
      o[g].x = (cr0 << 20) | (y0 << 10) | cb0;    // schrovirtframe.c:956-977 (pack_v210_s16)
      o[g].y = (y2 << 20) | (cb1 << 10) | y1;
      o[g].z = (cb2 << 20) | (y3 << 10) | cr1;
      o[g].w = (y5 << 20) | (cr2 << 10) | y4;
    }
    char *d = (char *) job.dst + (size_t) (8 * ty + y) * job.dst_stride + (size_t) (x0 / 6 + 2 * p) * 16;
    __builtin_nontemporal_store (o[0], (SCHRO_GLOBAL u32x4 *) d);
    __builtin_nontemporal_store (o[1], (SCHRO_GLOBAL u32x4 *) (d + 16));
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

// r05: which pictures the fused transform + v210 form takes: whole 48-pixel regions and 8-row strips, every coefficient
// row and band origin 16-byte aligned (the narrowest loads are the level-2 bands' single dwords; the level-0 bands' 16 bytes)
bool
iiwt_haar3_v210_ok (const HaarPackJob & j)
{
  uintptr_t bits = (uintptr_t) j.dst | (uintptr_t) j.dst_stride;
  for (int c = 0; c < 3; c++)
    bits |= (uintptr_t) j.src[c] | (uintptr_t) j.src_stride[c];
  return (bits & 15) == 0 && j.w % 48 == 0 && (j.w / 2) % 32 == 0 && j.h % 8 == 0 && j.w >= 48 && j.h >= 8;
}

int
iiwt_haar3_v210_strip_width ()
{
  return kHpWidth;
}

int
launch_iiwt_haar3_v210 (hipStream_t stream, const HaarPackJob * d_jobs, int njobs, int total_tiles, int filter)
{
  if (filter == 3)
    SCHRO_LAUNCH ((iiwt_haar3_v210_kernel < 0 >), dim3 (total_tiles), dim3 (kHaarThreads), 0, stream, d_jobs, njobs);
  else
    SCHRO_LAUNCH ((iiwt_haar3_v210_kernel < 1 >), dim3 (total_tiles), dim3 (kHaarThreads), 0, stream, d_jobs, njobs);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "iiwt (Haar s32, three levels + v210) launch: %s", hipGetErrorString (e));
  return 0;
}

}                               // namespace schro
