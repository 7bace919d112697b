// obmc_stage.hip -- OBMC + residual add + u8 clamp from LDS-staged reference windows.
//
// Same arithmetic as obmc.hip (schro_motion_render_u8, schromotion8.c:700-929; get_block
// :303-335; schro_upsampled_frame_get_block_fast_precN, schroframe.c:2288-2482;
// orc_rrshift6_add_s16_2d, schroorc.orc:636-661), default picture weights, half-pel
// references (mv_precision >= 1).  Different data flow:
//
// obmc.hip's item kernel GATHERS every (block, row) window from the tiled half-pel image
// through the vector L1: 26 M cache lines per 8 x 2160p launch for 129 MB of useful bytes,
// ~100 VALU lane-operations per output sample (r01 profiles).  Here a 512-thread workgroup
// owns a 128x32 output tile and, per reference,
//   1. decodes the tile's blocks once (one thread per block): get_block's clamped origin,
//      phases, and the bounding box of all sample windows of the tile's rows;
//   2. STREAMS that bounding box (one contiguous rectangle of the half-pel image, whole
//      128-byte lines, 8 lanes per line) into a linear LDS image;
//   3. predicts (block, row) items from LDS with byte-parallel arithmetic -- at quarter-pel
//      orc_combine4_nxm_u8 degenerates to copy / 2-sample / 4-sample rounding averages,
//      v_lerp_u8 on four pixels per instruction (exact: see avg4) -- items sorted by
//      (rx, ry) class so a wave's pass is straight-line code; result: one u8 row per item
//      in LDS, second reference blended in with avgub;
//   4. combines: a lane owns aligned 4-pixel groups, gathers the <= 4 x 2 covering blocks'
//      prediction bytes from LDS, multiplies by the separable OBMC ramp (v_pk_mad_u16: the
//      reference's s16 wrap), rounds, adds the residual, clamps, stores.
// Blocks whose window leaves the picture (per-sample clamp) or the staged rectangle (a
// vector far from the tile's other vectors) take an exact per-sample path from global
// memory into the same prediction rows; picture-edge weight folding (accumulate_slow,
// schromotion8.c:673-693) lives in the per-tile weight tables.
//
// Algorithmic bytes per output sample: as obmc.hip.  Bound: L2 -> LDS streaming of the
// windows (about 23 B per luma sample at +-16 pel vectors) overlapped with VALU.

#include "schro_hip_internal.h"
#include "obmc_common.h"
#include <algorithm>

namespace schro {
namespace {

constexpr int kSTW = 128, kSTH = 32;    // output tile (same as the item kernel: obmc_tiles)
constexpr int kSThreads = 512;
constexpr int kSWaves = kSThreads / 64;
constexpr int kStagePad = 64;           // bytes in front of / behind the staged image
constexpr int kStageBytes = 47616;      // 368-byte rows x 129 rows: a 128x32 luma tile at +-16 pel
constexpr int kStageLoads = 7;          // 16-byte chunks per thread and reference
constexpr int kMaxNCX = 31;             // widest staged row, chunks
constexpr int kPredBytes = 13824;       // prediction rows of one tile
constexpr int kSBlkCap = kSThreads;     // blocks per tile: one per thread
constexpr int kSItemCap = 2048 + 384;   // (block, row) items of one reference pass incl. class padding
constexpr int kNCls = 6;                // 0-3 (rx, ry) in {0,2}^2, 4 general bilinear, 5 exact per-sample
constexpr int kSGroupCap = 128;         // (class, block row) groups
constexpr int kSJCap = 21;              // block rows per tile

typedef unsigned short u16x2 __attribute__ ((ext_vector_type (2)));
typedef short s16x2 __attribute__ ((ext_vector_type (2)));

struct __attribute__ ((aligned (4))) SBlk {
  uint16_t wo[2];               // staged image: byte offset (+ kStagePad) of the first needed row's first sample
  uint8_t c[2];                 // per reference: kind << 4 | ry << 2 | rx; kind 0 fast, 1 exact, 3 unused
  uint8_t pl;                   // DC block: low byte of the prediction value
  int8_t ph;                    // DC block: the rest of it (0 in any legal stream)
  uint16_t pbase;               // first prediction row
  uint8_t nrows, ra;            // rows of the block inside the tile, first of them
};

struct __attribute__ ((aligned (16))) ColInfo {
  int a;                        // (block column * row pitch + (o & ~3)) << 2 | (o & 3): where the group's 4 bytes are
  uint32_t w01, w23;            // ramp weights of the 4 pixels (folded at picture edges), u16 each
  int irel;                     // block column in the tile (wide-DC correction only)
};

struct __attribute__ ((aligned (8))) RowInfo {
  int rowbase;                  // prediction row of block column 0
  uint32_t wy2;                 // vertical weight in both halves; 0: no block
};

__device__ __forceinline__ uint32_t
lerp1 (uint32_t a, uint32_t b)
{
  return __builtin_amdgcn_lerp (a, b, 0x01010101u);     // per byte (a + b + 1) >> 1 = avgub
}

// per byte (a + b + c + d + 2) >> 2, exactly: with c1 = (a+b+1)>>1, c2 = (c+d+1)>>1 the sum
// is 2 (c1 + c2) - l1 - l2 (l = the bit an average rounded up by), and
// (2 (c1 + c2 + 1) - l1 - l2) >> 2 = (c1 + c2 + 1 - (l1 | l2)) >> 1
__device__ __forceinline__ uint32_t
avg4 (uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
  const uint32_t t = (a ^ b) | (c ^ d);
  return __builtin_amdgcn_lerp (lerp1 (a, b), lerp1 (c, d), ~t);
}

__device__ __forceinline__ uint32_t
pk_mad (uint32_t a, uint32_t b, uint32_t c)
{
  return __builtin_bit_cast (uint32_t, (u16x2) (__builtin_bit_cast (u16x2, a) * __builtin_bit_cast (u16x2, b)
          + __builtin_bit_cast (u16x2, c)));
}

// wave-wide min / max by butterfly (the compiler turns a same-address LDS atomic min / max of
// 64 lanes into a 64-iteration scalar loop: 8 of them cost the decode phase 20 k cycles)
__device__ __forceinline__ int
wave_min (int v)
{
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
    v = min (v, __shfl_xor (v, d));
  return v;
}

__device__ __forceinline__ int
wave_max (int v)
{
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
    v = max (v, __shfl_xor (v, d));
  return v;
}

struct TileGeo {
  int x_lo, y_lo, x_hi, y_hi;
  int i_lo, j_lo, nbi, nbj, nblk;
  int rowpitch;                 // bytes between prediction rows: nbi * ppitch
};

// rows [ra, rb) of block row jj that lie in the tile
__device__ __forceinline__ void
block_rows (const ObmcJob & job, const TileGeo & tg, int jj, int *ra, int *rb)
{
  const int by = job.ybsep * jj - job.yoff;
  *ra = max (0, tg.y_lo - by);
  *rb = min (job.yblen, tg.y_hi - by);
}

// the predicted row of reference r for one item, ND dwords (4 pixels each), from the staged image
template < int ND, int CLS >
__device__ __forceinline__ void
predict_staged (const uint8_t * stage, int a, int pitch, int rx, int ry, uint32_t * out)
{
  const int sh = a & 3;
  const uint32_t *p0 = reinterpret_cast < const uint32_t * >(stage + (a & ~3));
  uint32_t d0[2 * ND + 1], e0[ND], o0[ND];
#pragma unroll
  for (int k = 0; k <= 2 * ND; k++)
    d0[k] = p0[k];
#pragma unroll
  for (int k = 0; k < ND; k++) {
    const uint32_t lo = __builtin_amdgcn_alignbyte (d0[2 * k + 1], d0[2 * k], sh);
    const uint32_t hi = __builtin_amdgcn_alignbyte (d0[2 * k + 2], d0[2 * k + 1], sh);
    e0[k] = __builtin_amdgcn_perm (hi, lo, 0x06040200u);       // half-pel columns hx + 2k
    if constexpr (CLS == 1 || CLS >= 3)
      o0[k] = __builtin_amdgcn_perm (hi, lo, 0x07050301u);     // hx + 1 + 2k
  }
  if constexpr (CLS == 0) {
#pragma unroll
    for (int k = 0; k < ND; k++)
      out[k] = e0[k];
    return;
  } else if constexpr (CLS == 1) {
#pragma unroll
    for (int k = 0; k < ND; k++)
      out[k] = lerp1 (e0[k], o0[k]);
    return;
  } else {
    const uint32_t *p1 = reinterpret_cast < const uint32_t * >(stage + (a & ~3) + pitch);
    uint32_t d1[2 * ND + 1], e1[ND], o1[ND];
#pragma unroll
    for (int k = 0; k <= 2 * ND; k++)
      d1[k] = p1[k];
#pragma unroll
    for (int k = 0; k < ND; k++) {
      const uint32_t lo = __builtin_amdgcn_alignbyte (d1[2 * k + 1], d1[2 * k], sh);
      const uint32_t hi = __builtin_amdgcn_alignbyte (d1[2 * k + 2], d1[2 * k + 1], sh);
      e1[k] = __builtin_amdgcn_perm (hi, lo, 0x06040200u);
      if constexpr (CLS >= 3)
        o1[k] = __builtin_amdgcn_perm (hi, lo, 0x07050301u);
    }
    if constexpr (CLS == 2) {
#pragma unroll
      for (int k = 0; k < ND; k++)
        out[k] = lerp1 (e0[k], e1[k]);
    } else if constexpr (CLS == 3) {
#pragma unroll
      for (int k = 0; k < ND; k++)
        out[k] = avg4 (e0[k], o0[k], e1[k], o1[k]);
    } else {
      // orc_combine4_nxm_u8 with any eighth-pel phase: (w00 a + w01 b + w10 c + w11 d + 8) >> 4
      const uint32_t wpk = (uint32_t) ((4 - ry) * (4 - rx)) | ((uint32_t) ((4 - ry) * rx) << 8)
          | ((uint32_t) (ry * (4 - rx)) << 16) | ((uint32_t) (ry * rx) << 24);
#pragma unroll
      for (int k = 0; k < ND; k++) {
        const uint32_t t0 = __builtin_amdgcn_perm (o0[k], e0[k], 0x05010400u);  // a0 b0 a1 b1
        const uint32_t t1 = __builtin_amdgcn_perm (o0[k], e0[k], 0x07030602u);  // a2 b2 a3 b3
        const uint32_t u0 = __builtin_amdgcn_perm (o1[k], e1[k], 0x05010400u);
        const uint32_t u1 = __builtin_amdgcn_perm (o1[k], e1[k], 0x07030602u);
        const uint32_t q0 = __builtin_amdgcn_udot4 (__builtin_amdgcn_perm (u0, t0, 0x05040100u), wpk, 8u, false) >> 4;
        const uint32_t q1 = __builtin_amdgcn_udot4 (__builtin_amdgcn_perm (u0, t0, 0x07060302u), wpk, 8u, false) >> 4;
        const uint32_t q2 = __builtin_amdgcn_udot4 (__builtin_amdgcn_perm (u1, t1, 0x05040100u), wpk, 8u, false) >> 4;
        const uint32_t q3 = __builtin_amdgcn_udot4 (__builtin_amdgcn_perm (u1, t1, 0x07060302u), wpk, 8u, false) >> 4;
        out[k] = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
      }
    }
  }
}

// exact per-sample path (window leaves the picture or the staged rectangle)
template < int ND >
__device__ __forceinline__ void
predict_exact (const ObmcJob & job, const TileGeo & tg, int b, int rr, int r, uint32_t * out)
{
  const int bj = b / tg.nbi, i = tg.i_lo + (b - bj * tg.nbi), jj = tg.j_lo + bj;
  const uint8_t *mvp = job.mvs + (size_t) 20 * ((size_t) jj * job.nbx + i);
  const uint32_t v01 = gload < uint32_t > (mvp + 12);
  const uint32_t v23 = gload < uint32_t > (mvp + 16);
  const int bx = job.xbsep * i - job.xoff, by = job.ybsep * jj - job.yoff;
  int fx, fy, ra, rb;
  mv_origin (job, bx, by, v01, v23, r, &fx, &fy);
  block_rows (job, tg, jj, &ra, &rb);
  const int row = ra + rr, prec = job.prec;
#pragma unroll
  for (int k = 0; k < ND; k++) {
    uint32_t v = 0;
    for (int e = 0; e < 4; e++) {
      const int x = 4 * k + e;
      if (x >= job.xblen)
        break;
      const int s = prec == 1
          ? fetch_ref < 1 > (job.ref[r], job.ref_stride[r], job.w, job.h, fx + x * 2, fy + row * 2, prec)
          : fetch_ref < 2 > (job.ref[r], job.ref_stride[r], job.w, job.h, fx + x * (1 << prec), fy + row * (1 << prec), prec);
      v |= (uint32_t) s << (8 * e);
    }
    out[k] = v;
  }
}

// scratch builds (SCHRO_HIP_OBMC_STAMPS): cycles since the workgroup started, per phase
#define STAMP(n) do { if (job.stamps && threadIdx.x == 0 && blockIdx.x < 16384) \
    job.stamps[blockIdx.x * 16 + (n)] = __builtin_amdgcn_s_memtime () - t_start; } while (0)

template < int ND >
__global__ __launch_bounds__ (kSThreads) __attribute__ ((amdgpu_waves_per_eu (4, 4)))
void obmc_stage_kernel (const ObmcJob * __restrict__ jobs, int njobs, const uint32_t * __restrict__ order)
{
  __shared__ __attribute__ ((aligned (16))) uint8_t s_stage[kStagePad + kStageBytes + kStagePad];
  __shared__ __attribute__ ((aligned (16))) uint8_t s_predbuf[16 + kPredBytes + 16];
  __shared__ SBlk s_blk[kSBlkCap];
  __shared__ uint16_t s_item[kSItemCap];
  __shared__ ColInfo s_col[kSTW / 4][4];
  __shared__ RowInfo s_row[kSTH][2];
  __shared__ int s_bbox[2][4];
  __shared__ int s_reg[2][8];           // X0 (chunks), Ymin, ncx, nry, pitch, valid, ny8, magic (ncx)
  __shared__ int s_gcnt[2][kSGroupCap];
  __shared__ uint16_t s_gstart[2][kSGroupCap];
  __shared__ int s_cstart[2][kNCls + 1], s_cend[2][kNCls];
  __shared__ int s_misc[4];             // 0: wide DC seen, 1: most block columns of a pixel group

  uint8_t *const s_pred = s_predbuf + 16;
  const uint64_t t_start = __builtin_amdgcn_s_memtime ();
  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  const uint32_t entry = order ? __builtin_amdgcn_readfirstlane (gload < uint32_t > (order + bid)) : 0u;
  const ObmcJob job = jobs[order ? (int) (entry >> 16) : find_job (jobs, njobs, bid)];
  const int t = order ? (int) (entry & 0xffffu) : bid - job.tile_base;
  const int ty = mdiv (t, job.tiles_x, job.m_tiles_x), tx = t - ty * job.tiles_x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xblen = job.xblen, yblen = job.yblen, xbsep = job.xbsep, ybsep = job.ybsep;
  const int xoff = job.xoff, yoff = job.yoff, prec = job.prec;
  constexpr int ppitch = 4 * ND;

  TileGeo tg;
  tg.x_lo = tx * kSTW;
  tg.y_lo = ty * kSTH;
  tg.x_hi = min (tg.x_lo + kSTW, job.w);
  tg.y_hi = min (tg.y_lo + kSTH, job.h);
  {
    // first / last block whose footprint meets the tile (numerators kept non-negative)
    tg.i_lo = max (0, mdiv (tg.x_lo + xoff - xblen + 2 * xbsep, xbsep, job.m_xbsep) - 1);
    const int i_hi = min (job.nbx - 1, mdiv (tg.x_hi - 1 + xoff, xbsep, job.m_xbsep));
    tg.j_lo = max (0, mdiv (tg.y_lo + yoff - yblen + 2 * ybsep, ybsep, job.m_ybsep) - 1);
    const int j_hi = min (job.nby - 1, mdiv (tg.y_hi - 1 + yoff, ybsep, job.m_ybsep));
    tg.nbi = i_hi - tg.i_lo + 1;
    tg.nbj = j_hi - tg.j_lo + 1;
    tg.nblk = tg.nbi * tg.nbj;
    tg.rowpitch = tg.nbi * ppitch;
  }
  const int gw = 2 * job.w - 2, gh = 2 * job.h - 2;     // last valid half-pel sample column / row

  // ---- this lane's output pixels (combine phase): group gq of rows ys and ys + 16; the
  // residual is fetched now and used at the very end --------------------------------------
  const int gq = tid & 31, ys = tid >> 5;
  const int ox = tg.x_lo + 4 * gq;
  const bool res_fast = job.res_bpp == 2 && ((((uintptr_t) job.residual) | (uintptr_t) job.residual_stride) & 7) == 0;
  uint32_t res_lo[2] = { 0u, 0u }, res_hi[2] = { 0u, 0u };
  if (res_fast && ox + 4 <= job.w) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int y = tg.y_lo + ys + 16 * k;
      if (y < tg.y_hi) {
        const u32x2 rv = gload < u32x2 > ((const char *) job.residual + (size_t) y * job.residual_stride + 2 * ox);
        res_lo[k] = rv.x;
        res_hi[k] = rv.y;
      }
    }
  }

  if (tid < 8)
    s_bbox[tid >> 2][tid & 3] = (tid & 1) ? -0x40000000 : 0x40000000;   // xmin xmax ymin ymax
  if (tid < 2 * kSGroupCap)
    s_gcnt[tid / kSGroupCap][tid % kSGroupCap] = 0;
  if (tid < 4)
    s_misc[tid] = 0;

  // ---- weight tables ---------------------------------------------------------------------
  if (tid < kSTW / 4) {
    // the <= 4 block columns that cover pixel group gq = tid and their (folded) ramp weights
    const int g = tid, x0 = tg.x_lo + 4 * g;
    int imin = 0x7fffffff, bi[4][2], wv[4][2];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int px = x0 + e, u = px + xoff;
      const int i0 = mdiv (u, xbsep, job.m_xbsep), r = u - i0 * xbsep;
      int wa = 0, wb = 0;
      const bool has_a = (r < 2 * xoff) && (i0 - 1 >= 0) && (i0 - 1 < job.nbx);
      const bool has_b = (i0 < job.nbx);
      if (r < 2 * xoff) {
        wa = obmc_weight_1d (r + xbsep, xblen, xoff);
        wb = obmc_weight_1d (r, xblen, xoff);
        if (!has_a)
          wb += wa;
        if (!has_b)
          wa += wb;
      } else {
        wb = 8;
      }
      const bool in = px < job.w;
      bi[e][0] = in && has_a ? i0 - 1 : -1;
      wv[e][0] = wa;
      bi[e][1] = in && has_b ? i0 : -1;
      wv[e][1] = wb;
      if (bi[e][0] >= 0)
        imin = min (imin, bi[e][0]);
      if (bi[e][1] >= 0)
        imin = min (imin, bi[e][1]);
    }
    if (imin == 0x7fffffff)
      imin = 0;                 // group outside the picture: no weights at all
    int nc = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int i = imin + c;
      uint32_t w[4] = { 0, 0, 0, 0 };
      bool any = false;
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int s = 0; s < 2; s++)
          if (bi[e][s] == i) {
            w[e] = (uint32_t) wv[e][s];
            any = true;
          }
      ColInfo ci;
      const int irel = any ? i - tg.i_lo : 0;
      const int o = any ? x0 - (xbsep * i - xoff) : 0;
      ci.a = ((irel * ppitch + (o & ~3)) << 2) | (o & 3);
      ci.w01 = w[0] | (w[1] << 16);
      ci.w23 = w[2] | (w[3] << 16);
      ci.irel = irel;
      s_col[g][c] = ci;
      if (any)
        nc = c + 1;
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1)   // the 32 lanes of this branch
      nc = max (nc, __shfl_xor (nc, d));
    if (tid == 0)
      s_misc[1] = nc;
  } else if (tid >= 64 && tid < 64 + kSTH) {
    const int yy = tid - 64, y = tg.y_lo + yy, u = y + yoff;
    const int j0 = mdiv (u, ybsep, job.m_ybsep), r = u - j0 * ybsep;
    int wa = 0, wb = 0;
    const bool has_a = (r < 2 * yoff) && (j0 - 1 >= 0) && (j0 - 1 < job.nby);
    const bool has_b = (j0 < job.nby);
    if (r < 2 * yoff) {
      wa = obmc_weight_1d (r + ybsep, yblen, yoff);
      wb = obmc_weight_1d (r, yblen, yoff);
      if (!has_a)
        wb += wa;
      if (!has_b)
        wa += wb;
    } else {
      wb = 8;
    }
    const int jsel[2] = { has_a ? j0 - 1 : -1, has_b ? j0 : -1 };
    const int wsel[2] = { wa, wb };
#pragma unroll
    for (int v = 0; v < 2; v++) {
      RowInfo ri;
      ri.rowbase = 0;
      ri.wy2 = 0;
      if (jsel[v] >= 0 && y < tg.y_hi) {
        // prediction rows are laid out block row after block row, the rows inside the tile only
        int slot = 0;
        for (int jj = tg.j_lo; jj < jsel[v]; jj++) {
          int ra, rb;
          block_rows (job, tg, jj, &ra, &rb);
          slot += rb - ra;
        }
        int ra, rb;
        block_rows (job, tg, jsel[v], &ra, &rb);
        slot += (y - (ybsep * jsel[v] - yoff)) - ra;
        ri.rowbase = slot * tg.rowpitch;
        ri.wy2 = (uint32_t) wsel[v] * 0x10001u;
      }
      s_row[yy][v] = ri;
    }
  }
  __syncthreads ();

  STAMP (1);
  // ---- decode: one block per thread -------------------------------------------------------
  const bool have = tid < tg.nblk;
  int my_j = 0, my_nrows = 0, my_ra = 0, my_pbase = 0;
  int my_gx[2] = { 0, 0 }, my_gy[2] = { 0, 0 };        // window origin of the first needed row
  int my_c[2] = { 0x30, 0x30 }, my_pl = 0, my_ph = 0;
  bool my_inpic[2] = { false, false };
  int my_x0[2] = { 0, 0 }, my_x1[2] = { 0, 0 }, my_y1[2] = { 0, 0 };
  if (have) {
    const int bj = tid / tg.nbi;
    const int i = tg.i_lo + (tid - bj * tg.nbi), jj = tg.j_lo + bj;
    my_j = bj;
    const uint8_t *mvp = job.mvs + (size_t) 20 * ((size_t) jj * job.nbx + i);
    const uint32_t flags = gload < uint32_t > (mvp);
    const uint32_t v01 = gload < uint32_t > (mvp + 12);
    const uint32_t v23 = gload < uint32_t > (mvp + 16);
    const int bx = xbsep * i - xoff, by = ybsep * jj - yoff;
    int ra, rb;
    block_rows (job, tg, jj, &ra, &rb);
    my_ra = ra;
    my_nrows = rb - ra;
    int slot = 0;
    for (int j2 = tg.j_lo; j2 < jj; j2++) {
      int a2, b2;
      block_rows (job, tg, j2, &a2, &b2);
      slot += b2 - a2;
    }
    my_pbase = slot * tg.rowpitch + (tid - bj * tg.nbi) * ppitch;
    const int mode = flags & 3;
    if (mode == 0) {
      const bool interior = i >= 1 && i < job.max_x_blocks && jj >= 1 && jj < job.max_y_blocks;
      const int dc = job.comp == 0 ? (int16_t) (v01 & 0xffff)
          : job.comp == 1 ? (int16_t) (v01 >> 16) : (int16_t) (v23 & 0xffff);
      // get_dc_block stores a uint8_t; block_acc_dc multiplies a 16-bit parameter
      const int p = interior ? (int) (int16_t) (dc + 128) : (int) (uint8_t) (dc + 128);
      my_pl = p & 255;
      my_ph = p >> 8;
      if (my_ph)
        s_misc[0] = 1;
      const uint32_t fill = (uint32_t) my_pl * 0x01010101u;
      for (int rr = 0; rr < my_nrows; rr++)
#pragma unroll
        for (int k = 0; k < ND; k++)
          *reinterpret_cast < uint32_t * >(s_pred + my_pbase + rr * tg.rowpitch + 4 * k) = fill;
    }
    // columns of the block inside the tile
    const int ca = max (0, tg.x_lo - bx), cb = min (xblen, tg.x_hi - bx);
#pragma unroll
    for (int r = 0; r < 2; r++) {
      if (!(mode & (r + 1)))
        continue;
      int fx, fy;
      mv_origin (job, bx, by, v01, v23, r, &fx, &fy);
      int gx0, gy0, rx = 0, ry = 0;
      if (prec == 1) {
        gx0 = fx;
        gy0 = fy;
      } else {
        const int x8 = prec == 2 ? fx * 2 : fx, y8 = prec == 2 ? fy * 2 : fy;
        rx = x8 & 3;
        ry = y8 & 3;
        gx0 = x8 >> 2;
        gy0 = y8 >> 2;
      }
      gy0 += 2 * ra;
      my_gx[r] = gx0;
      my_gy[r] = gy0;
      my_c[r] = (ry << 2) | rx;
      // the samples this tile needs of the block: columns [ca, cb), rows [ra, rb), both bilinear taps
      const int wx0 = gx0 + 2 * ca, wx1 = gx0 + 2 * cb - 1, wy1 = gy0 + 2 * (my_nrows - 1) + 1;
      my_x0[r] = wx0;
      my_x1[r] = wx1;
      my_inpic[r] = wx0 >= 0 && gy0 >= 0 && wx1 <= gw && wy1 <= gh;
      my_y1[r] = wy1;
    }
  }
  // bounding box of the windows that need no clamping, per reference: per wave, then one lane
  if (wave * 64 < tg.nblk) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const bool in = my_inpic[r];
      const int x0 = wave_min (in ? my_x0[r] : 0x40000000), x1 = wave_max (in ? my_x1[r] : -0x40000000);
      const int y0 = wave_min (in ? my_gy[r] : 0x40000000), y1 = wave_max (in ? my_y1[r] : -0x40000000);
      if (lane == 0 && x0 <= x1) {
        atomicMin (&s_bbox[r][0], x0);
        atomicMax (&s_bbox[r][1], x1);
        atomicMin (&s_bbox[r][2], y0);
        atomicMax (&s_bbox[r][3], y1);
      }
    }
  }
  __syncthreads ();

  STAMP (2);
  // ---- the staged rectangle of each reference ---------------------------------------------
  if (tid < 2) {
    const int r = tid;
    const int xmin = s_bbox[r][0], xmax = s_bbox[r][1];
    int ymin = s_bbox[r][2];
    const int ymax = s_bbox[r][3];
    int valid = xmin <= xmax, X0 = 0, ncx = 1, nry = 0, pitch = 16;
    if (valid) {
      X0 = xmin >> 4;
      ncx = (xmax >> 4) - X0 + 1;
      if (ncx > kMaxNCX) {      // vectors too far apart: keep the middle, the rest goes the exact way
        X0 += (ncx - kMaxNCX) >> 1;
        ncx = kMaxNCX;
      }
      pitch = 16 * (ncx | 1);   // odd number of 16-byte units: 8 rows of a cache line hit 8 different bank groups
      nry = ymax - ymin + 1;
      const int ny8max = (kStageLoads * kSThreads) / (8 * ncx);
      // (a band of 16 rows = two tile rows; the window's rows can start anywhere in a band)
      const int maxrows = min (kStageBytes / pitch, 16 * ((ny8max >> 1) - 1));
      if (nry > maxrows) {
        ymin += (nry - maxrows) >> 1;
        nry = maxrows;
      }
    }
    s_reg[r][0] = X0;
    s_reg[r][1] = ymin;
    s_reg[r][2] = ncx;
    s_reg[r][3] = nry;
    s_reg[r][4] = pitch;
    s_reg[r][5] = valid;
    s_reg[r][6] = valid ? 2 * (((ymin + nry - 1) >> 4) - (ymin >> 4) + 1) : 0;        // tile rows (8 rows of one parity)
    s_reg[r][7] = (int) div_magic (ncx);
  }
  __syncthreads ();

  // stage loads of one reference: 8 consecutive lanes fetch the 8 rows of one 128-byte line
  auto stage_load = [&] (int r, u32x4 * v) {
    const int X0 = s_reg[r][0], ymin = s_reg[r][1], ncx = s_reg[r][2], nry = s_reg[r][3];
    const int nchunk = 8 * ncx * s_reg[r][6];
    const uint32_t m = (uint32_t) s_reg[r][7];
    const int ymax_mem = 2 * job.h - 1, cmax_mem = (job.ref_stride[r] >> 4) - 1;
#pragma unroll
    for (int k = 0; k < kStageLoads; k++) {
      const int idx = tid + k * kSThreads;
      if (idx < nchunk) {
        const int q = idx >> 3, y8 = mdiv (q, ncx, m), cx = q - y8 * ncx;
        const int yy = (ymin & ~15) + 16 * (y8 >> 1) + (y8 & 1) + 2 * (idx & 7);
        if (yy >= ymin && yy < ymin + nry) {
          const int ym = min (yy, ymax_mem), cm = min (X0 + cx, cmax_mem);
          v[k] = gload < u32x4 > (job.ref[r] + hp_row_offset (ym, job.ref_stride[r]) + (size_t) cm * 128);
        }
      }
    }
  };
  auto stage_store = [&] (int r, const u32x4 * v) {
    const int ymin = s_reg[r][1], ncx = s_reg[r][2], nry = s_reg[r][3], pitch = s_reg[r][4];
    const int nchunk = 8 * ncx * s_reg[r][6];
    const uint32_t m = (uint32_t) s_reg[r][7];
#pragma unroll
    for (int k = 0; k < kStageLoads; k++) {
      const int idx = tid + k * kSThreads;
      if (idx < nchunk) {
        const int q = idx >> 3, y8 = mdiv (q, ncx, m), cx = q - y8 * ncx;
        const int yy = (ymin & ~15) + 16 * (y8 >> 1) + (y8 & 1) + 2 * (idx & 7);
        if (yy >= ymin && yy < ymin + nry)
          *reinterpret_cast < u32x4 * >(s_stage + kStagePad + (yy - ymin) * pitch + 16 * cx) = v[k];
      }
    }
  };

  u32x4 sv[kStageLoads];
  if (s_reg[0][5])
    stage_load (0, sv);

  // ---- classify against the rectangles; count the (class, block row) groups ----------------
  int my_rank[2] = { 0, 0 }, my_grp[2] = { 0, 0 };
  if (have) {
    SBlk sb;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      sb.wo[r] = 0;
      if ((my_c[r] >> 4) == 3)
        continue;
      const int X0 = s_reg[r][0] * 16, ymin = s_reg[r][1], ncx = s_reg[r][2], nry = s_reg[r][3], pitch = s_reg[r][4];
      const bool fast = my_inpic[r] && s_reg[r][5] && my_x0[r] >= X0 && my_x1[r] < X0 + 16 * ncx
          && my_gy[r] >= ymin && my_gy[r] + 2 * (my_nrows - 1) + 1 < ymin + nry;
      int cls;
      if (fast) {
        sb.wo[r] = (uint16_t) (kStagePad + (my_gy[r] - ymin) * pitch + (my_gx[r] - X0));
        const int rx = my_c[r] & 3, ry = my_c[r] >> 2;
        cls = ((rx | ry) & 1) ? 4 : ((rx >> 1) | (ry & 2));
      } else {
        my_c[r] |= 0x10;
        cls = 5;
      }
      my_grp[r] = cls * tg.nbj + my_j;
      my_rank[r] = atomicAdd (&s_gcnt[r][my_grp[r]], 1);
    }
    sb.c[0] = (uint8_t) my_c[0];
    sb.c[1] = (uint8_t) my_c[1];
    sb.pl = (uint8_t) my_pl;
    sb.ph = (int8_t) my_ph;
    sb.pbase = (uint16_t) my_pbase;
    sb.nrows = (uint8_t) my_nrows;
    sb.ra = (uint8_t) my_ra;
    s_blk[tid] = sb;
  }
  __syncthreads ();

  STAMP (3);
  // ---- item ranges: groups of one class follow each other, classes start on a multiple of 64 so
  // that the 64 items of a wave's pass are one class ----------------------------------------
  if (wave < 2) {
    const int r = wave;
    int rows_l = 0;
    if (lane < tg.nbj) {
      int ra, rb;
      block_rows (job, tg, tg.j_lo + lane, &ra, &rb);
      rows_l = rb - ra;
    }
    int base = 0;
    for (int c = 0; c < kNCls; c++) {
      const int v = lane < tg.nbj ? s_gcnt[r][c * tg.nbj + lane] * rows_l : 0;
      int incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up (incl, d);
        if (lane >= d)
          incl += up;
      }
      if (lane < tg.nbj)
        s_gstart[r][c * tg.nbj + lane] = (uint16_t) (base + incl - v);
      const int total = __shfl (incl, 63);
      if (lane == 0) {
        s_cstart[r][c] = base;
        s_cend[r][c] = base + total;
      }
      base = (base + total + 63) & ~63;
    }
    if (lane == 0)
      s_cstart[r][kNCls] = base;
  }
  __syncthreads ();

  auto write_items = [&] (int r) {
    if (have && (my_c[r] >> 4) != 3) {
      const int start = s_gstart[r][my_grp[r]] + my_rank[r] * my_nrows;
      for (int rr = 0; rr < my_nrows; rr++)
        s_item[start + rr] = (uint16_t) (tid | (rr << 9));
    }
  };

  // one reference pass: every wave takes 64 items of one class at a time
  auto predict_pass = [&] (int r) {
    const int pitch = s_reg[r][4];
    const int nchunks = s_cstart[r][kNCls] >> 6;
    for (int q = wave; q < nchunks; q += kSWaves) {
      const int base = q << 6;
      int cls = 0;
#pragma unroll
      for (int c = 1; c < kNCls; c++)
        cls += base >= s_cstart[r][c];
      const int it = base + lane;
      const bool valid = it < s_cend[r][cls];
      if (!valid)
        continue;
      const int e = s_item[it];
      const int b = e & 511, rr = e >> 9;
      const SBlk sb = s_blk[b];
      const int a = (int) sb.wo[r] + 2 * rr * pitch;
      const int rx = sb.c[r] & 3, ry = (sb.c[r] >> 2) & 3;
      uint32_t out[ND];
      switch (cls) {
        case 0: predict_staged < ND, 0 > (s_stage, a, pitch, rx, ry, out); break;
        case 1: predict_staged < ND, 1 > (s_stage, a, pitch, rx, ry, out); break;
        case 2: predict_staged < ND, 2 > (s_stage, a, pitch, rx, ry, out); break;
        case 3: predict_staged < ND, 3 > (s_stage, a, pitch, rx, ry, out); break;
        case 4: predict_staged < ND, 4 > (s_stage, a, pitch, rx, ry, out); break;
        default: predict_exact < ND > (job, tg, b, rr, r, out); break;
      }
      uint32_t *pp = reinterpret_cast < uint32_t * >(s_pred + sb.pbase + rr * tg.rowpitch);
      if (r == 1 && (sb.c[0] >> 4) != 3) {
        // both references: avgub of the two predictions (schromotion8.c:560-566 with the default weights)
#pragma unroll
        for (int k = 0; k < ND; k++)
          out[k] = lerp1 (pp[k], out[k]);
      }
#pragma unroll
      for (int k = 0; k < ND; k++)
        pp[k] = out[k];
    }
  };

  STAMP (4);
  write_items (0);
  if (s_reg[0][5])
    stage_store (0, sv);
  __syncthreads ();
  STAMP (5);
  // the second reference's window travels while the first one is being worked on
  if (s_reg[1][5])
    stage_load (1, sv);
  predict_pass (0);
  __syncthreads ();
  STAMP (6);
  write_items (1);
  if (s_reg[1][5])
    stage_store (1, sv);
  __syncthreads ();
  STAMP (7);
  predict_pass (1);
  __syncthreads ();
  STAMP (8);

  // ---- combine: sum over the covering blocks, round, add the residual, clamp ---------------
  if (ox >= job.w)
    return;
  const bool stamp_end = job.stamps && threadIdx.x == 0 && blockIdx.x < 16384;
  const int ncmax = s_misc[1];
  const bool wide = s_misc[0] != 0;
  ColInfo ci[4];
#pragma unroll
  for (int c = 0; c < 4; c++)
    ci[c] = s_col[gq][c];
  const bool out_fast = ox + 4 <= job.w && ((((uintptr_t) job.out) | (uintptr_t) job.out_stride) & 3) == 0;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int yy = ys + 16 * k, y = tg.y_lo + yy;
    if (y >= tg.y_hi)
      continue;
    uint32_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int v = 0; v < 2; v++) {
      const RowInfo ri = s_row[yy][v];
      if (ri.wy2 == 0)
        continue;
      uint32_t ax0 = 0, ax1 = 0;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        if (c >= ncmax)
          break;
        const int addr = ri.rowbase + (ci[c].a >> 2);
        const uint32_t lo = *reinterpret_cast < const uint32_t * >(s_pred + addr);
        const uint32_t hi = *reinterpret_cast < const uint32_t * >(s_pred + addr + 4);
        const uint32_t pv = __builtin_amdgcn_alignbyte (hi, lo, (uint32_t) ci[c].a & 3u);
        ax0 = pk_mad (__builtin_amdgcn_perm (0u, pv, 0x0c010c00u), ci[c].w01, ax0);
        ax1 = pk_mad (__builtin_amdgcn_perm (0u, pv, 0x0c030c02u), ci[c].w23, ax1);
        if (wide) {
          // a DC value outside 0..255 (no legal stream): the prediction rows hold its low byte,
          // the rest enters the 16-bit sums here
          const int brow = ri.rowbase / tg.rowpitch;    // prediction row -> block row: search
          int slot = 0, bj = 0;
          for (int jj = tg.j_lo; jj < tg.j_lo + tg.nbj; jj++) {
            int ra, rb;
            block_rows (job, tg, jj, &ra, &rb);
            if (brow < slot + rb - ra)
              break;
            slot += rb - ra;
            bj++;
          }
          const SBlk sb = s_blk[bj * tg.nbi + ci[c].irel];
          const uint32_t hv = (uint32_t) ((((sb.c[0] & sb.c[1]) >> 4) == 3 ? (int) sb.ph : 0) << 8) & 0xffffu;
          const uint32_t h2 = hv * 0x10001u;
          if (ci[c].w01 | ci[c].w23) {
            ax0 = pk_mad (h2, ci[c].w01, ax0);
            ax1 = pk_mad (h2, ci[c].w23, ax1);
          }
        }
      }
      acc0 = pk_mad (ax0, ri.wy2, acc0);
      acc1 = pk_mad (ax1, ri.wy2, acc1);
    }
    // orc_rrshift6_add_s16_2d / _s32_2d: 16-bit wrapping arithmetic throughout
    s16x2 t0 = (__builtin_bit_cast (s16x2, acc0) + (short) 32) >> 6;
    s16x2 t1 = (__builtin_bit_cast (s16x2, acc1) + (short) 32) >> 6;
    const char *rrow = (const char *) job.residual + (size_t) y * job.residual_stride;
    uint8_t *orow = job.out + (size_t) y * job.out_stride + ox;
    s16x2 r0, r1;
    if (res_fast && ox + 4 <= job.w) {
      r0 = __builtin_bit_cast (s16x2, res_lo[k]);
      r1 = __builtin_bit_cast (s16x2, res_hi[k]);
    } else {
      short rv[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        rv[e] = 0;
        if (ox + e < job.w)
          rv[e] = job.res_bpp == 2 ? gload < int16_t > ((const int16_t *) rrow + ox + e)
              : (int16_t) gload < int32_t > ((const int32_t *) rrow + ox + e);  // convlw
      }
      r0 = (s16x2) { rv[0], rv[1] };
      r1 = (s16x2) { rv[2], rv[3] };
    }
    t0 = t0 + r0;
    t1 = t1 + r1;
    t0 = __builtin_elementwise_min (__builtin_elementwise_max (t0, (s16x2) (short) 0), (s16x2) (short) 255);
    t1 = __builtin_elementwise_min (__builtin_elementwise_max (t1, (s16x2) (short) 0), (s16x2) (short) 255);
    const uint32_t pk = __builtin_amdgcn_perm (__builtin_bit_cast (uint32_t, t1), __builtin_bit_cast (uint32_t, t0), 0x06040200u);
    if (out_fast) {
      gstore < uint32_t > (orow, pk);
    } else {
      for (int e = 0; e < 4 && ox + e < job.w; e++)
        gstore < uint8_t > (orow + e, (uint8_t) (pk >> (8 * e)));
    }
  }
  if (stamp_end)
    job.stamps[blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memtime () - t_start;
}

template < int ND >
int
launch_nd (hipStream_t stream, const ObmcJob * d_jobs, int njobs, int total_tiles, const uint32_t * d_order)
{
  hipLaunchKernelGGL ((obmc_stage_kernel < ND >), dim3 (total_tiles), dim3 (kSThreads), 0, stream, d_jobs, njobs, d_order);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "obmc (staged) launch: %s", hipGetErrorString (e));
  return 0;
}

}                               // namespace

// Prediction dwords per block row the staged kernel would run this plane with, or 0 when the
// plane's geometry does not fit its fixed LDS tables (then the item kernel of obmc.hip runs).
int
obmc_stage_nd (const ObmcJob & j)
{
  if (j.prec < 1 || j.xbsep < 2 || j.ybsep < 2)
    return 0;
  if ((((uintptr_t) j.ref[0]) | ((uintptr_t) j.ref[1]) | (uintptr_t) j.ref_stride[0] | (uintptr_t) j.ref_stride[1]) & 15)
    return 0;
  const int need = (j.xblen + 3) / 4;
  const int nd = need <= 2 ? 2 : need <= 3 ? 3 : need <= 4 ? 4 : need <= 6 ? 6 : 0;
  if (!nd)
    return 0;
  const int nbi = (kSTW + j.xblen - 2) / j.xbsep + 2, nbj = (kSTH + j.yblen - 2) / j.ybsep + 2;
  const int rowslots = kSTH + (kSTH / j.ybsep + 2) * 2 * j.yoff;
  if (nbi * nbj > kSBlkCap || nbj > kSJCap || kNCls * nbj > kSGroupCap || j.yblen > 63)
    return 0;
  if (rowslots * nbi * 4 * nd > kPredBytes || rowslots * nbi + kNCls * 64 > kSItemCap)
    return 0;
  return nd;
}

int
launch_obmc_stage (hipStream_t stream, const ObmcJob * d_jobs, int njobs, int total_tiles, int nd,
    const uint32_t * d_order)
{
  switch (nd) {
    case 2: return launch_nd < 2 > (stream, d_jobs, njobs, total_tiles, d_order);
    case 3: return launch_nd < 3 > (stream, d_jobs, njobs, total_tiles, d_order);
    case 4: return launch_nd < 4 > (stream, d_jobs, njobs, total_tiles, d_order);
    case 6: return launch_nd < 6 > (stream, d_jobs, njobs, total_tiles, d_order);
  }
  return set_error (SCHRO_HIP_EINVAL, "obmc (staged): %d dwords per row unsupported", nd);
}

}                               // namespace schro
