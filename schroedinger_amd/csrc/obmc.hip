// obmc.hip -- overlapped block motion compensation + residual add + u8 clamp.
//
// What it computes: schro_motion_render (motion, dest, addframe, add=TRUE,
// output_frame) -> schro_motion_render_u8 (schroedinger/schromotion8.c:700-929)
// for one component: for every block (i,j) a predicted xblen x yblen block
// (DC, one reference, or two references; sub-pel fetch get_block :303-335 ->
// schro_upsampled_frame_get_block_fast_precN schroframe.c:2459-2482), weighted
// by the separable OBMC ramp (schromotion.c:40-93), accumulated in s16, then
// out = sat_u8 (residual + ((acc + 32) >> 6)) (orc_rrshift6_add_s16_2d,
// schroorc.orc:636-661).
//
// How (MI355X-first): the reference SCATTERS blocks into an s16 frame and
// finalises block rows; every add is a 16-bit wrapping add, so the order is
// irrelevant and the same value is obtained by a per-pixel GATHER over the
// <= 2x2 blocks covering the pixel.  One thread = one output pixel; no s16
// accumulator frame exists in memory at all (it stays in a register), the
// residual is read once and the u8 written once.  Details kept bit-exact:
//   * edge blocks (i == 0, j == 0, i >= max_x_blocks, j >= max_y_blocks) use
//     the u8 "predict_block" arithmetic, interior blocks the s16 Orc-program
//     arithmetic (schromotion8.c:542-657) -- they differ for weight gain > 1;
//   * at picture edges the weight of the missing neighbour block is folded
//     into the existing one (accumulate_slow :673-693);
//   * the block position is clamped like get_block :329-330, and each sample
//     coordinate is clamped to the half-pel image, which is what the
//     reference's 32-pixel aprons hold (schroframe.c:1940-2030).
// References are the interleaved half-pel images written by upsample_kernel
// (mv_precision >= 1) or plain u8 planes (mv_precision == 0).
//
// Bound: HBM/L2 gather.  Algorithmic bytes per output sample: residual 2|4 B
// + output 1 B + 1 B per reference used.

#include "schro_hip_internal.h"
#include "obmc_common.h"
#include <algorithm>

namespace schro {
namespace {

constexpr int kThreads = 256;
constexpr int kTW = 64, kTH = 4;

template < int PC, bool SIMPLE >
__global__ __launch_bounds__ (kThreads)
void obmc_kernel (const ObmcJob * __restrict__ jobs, int njobs, uint32_t * __restrict__ overflow)
{
  __shared__ int s_wx[kMaxBlk], s_wy[kMaxBlk];

  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  const ObmcJob job = jobs[find_job (jobs, njobs, bid)];
  const int t = bid - job.tile_base;
  const int tx = t % job.tiles_x, ty = t / job.tiles_x;
  const int tid = threadIdx.x;

  if (tid < job.xblen)
    s_wx[tid] = obmc_weight_1d (tid, job.xblen, job.xoff);
  if (tid >= 64 && tid - 64 < job.yblen)
    s_wy[tid - 64] = obmc_weight_1d (tid - 64, job.yblen, job.yoff);
  __syncthreads ();

  const int px = tx * kTW + (tid % kTW);
  const int py = ty * kTH + (tid / kTW);
  if (px >= job.w || py >= job.h)
    return;

  // blocks covering this pixel in x: i0 (index rx) and, inside the ramp,
  // i0 - 1 (index rx + xbsep); a missing one folds its weight into the other
  int bi[2], wxs[2], nx = 0;
  {
    int u = px + job.xoff;
    int i0 = u / job.xbsep, r = u - i0 * job.xbsep;
    int wa = 0, wb = 0;
    bool has_a = (r < 2 * job.xoff) && (i0 - 1 >= 0) && (i0 - 1 < job.nbx);
    bool has_b = (i0 < job.nbx);
    if (r < 2 * job.xoff) {
      wa = s_wx[r + job.xbsep];
      wb = s_wx[r];
      if (!has_a) { wb += wa; }
      if (!has_b) { wa += wb; }
    } else {
      wb = 8;
    }
    if (has_a) { bi[nx] = i0 - 1; wxs[nx] = wa; nx++; }
    if (has_b) { bi[nx] = i0; wxs[nx] = wb; nx++; }
  }
  int bj[2], wys[2], ny = 0;
  {
    int u = py + job.yoff;
    int j0 = u / job.ybsep, r = u - j0 * job.ybsep;
    int wa = 0, wb = 0;
    bool has_a = (r < 2 * job.yoff) && (j0 - 1 >= 0) && (j0 - 1 < job.nby);
    bool has_b = (j0 < job.nby);
    if (r < 2 * job.yoff) {
      wa = s_wy[r + job.ybsep];
      wb = s_wy[r];
      if (!has_a) { wb += wa; }
      if (!has_b) { wa += wb; }
    } else {
      wb = 8;
    }
    if (has_a) { bj[ny] = j0 - 1; wys[ny] = wa; ny++; }
    if (has_b) { bj[ny] = j0; wys[ny] = wb; ny++; }
  }

  const int prec = job.prec;
  const int expx = 32 << prec;
  const int max_fast_x = (job.w - job.xblen) * (1 << prec);
  const int max_fast_y = (job.h - job.yblen) * (1 << prec);
  const int wsum = job.w1 + job.w2;
  const bool noscale = (wsum == (1 << job.wbits));

  int acc = 0;                  // s16 accumulator, kept modulo 2^16
  for (int b = 0; b < ny; b++) {
    for (int a = 0; a < nx; a++) {
      const int i = bi[a], jj = bj[b];
      const uint8_t *mvp = job.mvs + (size_t) 20 * ((size_t) jj * job.nbx + i);
      const uint32_t flags = gload < uint32_t > (mvp);
      const uint32_t v01 = gload < uint32_t > (mvp + 12);
      const uint32_t v23 = gload < uint32_t > (mvp + 16);
      const int mode = flags & 3;
      const bool interior = i >= 1 && i < job.max_x_blocks && jj >= 1 && jj < job.max_y_blocks;
      const int bx = job.xbsep * i - job.xoff, by = job.ybsep * jj - job.yoff;
      const int wgt = wxs[a] * wys[b];
      int pred;                 // value that gets multiplied by the OBMC weight

      if (mode == 0) {
        int dc = job.comp == 0 ? (int16_t) (v01 & 0xffff)
            : job.comp == 1 ? (int16_t) (v01 >> 16) : (int16_t) (v23 & 0xffff);
        // get_dc_block stores into a uint8_t; block_acc_dc multiplies a 16-bit parameter
        pred = interior ? (int) (int16_t) (dc + 128) : (int) (uint8_t) (dc + 128);
        if (overflow && (unsigned) pred > 255u)         // (prediction_only launches: see obmc_row.hip)
          __hip_atomic_store (overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        int val[2] = { 0, 0 };
#pragma unroll
        for (int r = 0; r < 2; r++) {
          if (!(mode & (r + 1)))
            continue;
          int dx = r == 0 ? (int16_t) (v01 & 0xffff) : (int16_t) (v01 >> 16);
          int dy = r == 0 ? (int16_t) (v23 & 0xffff) : (int16_t) (v23 >> 16);
          dx >>= job.mv_shift_x;
          dy >>= job.mv_shift_y;
          int fx = clampi (bx * (1 << prec) + dx, -expx, max_fast_x + expx - 1);
          int fy = clampi (by * (1 << prec) + dy, -expx, max_fast_y + expx - 1);
          int sx = fx + (px - bx) * (1 << prec);
          int sy = fy + (py - by) * (1 << prec);
          val[r] = fetch_ref < PC > (job.ref[r], job.ref_stride[r], job.w, job.h, sx, sy, prec, job.ref_ps, job.ref_cb);
        }
        if (mode == 3) {
          if constexpr (SIMPLE) {
            pred = (val[0] + val[1] + 1) >> 1;  // avgub, both paths
          } else if (interior) {
            // block_acc_biref, schromotion8.c:131-163
            int16_t t1 = (int16_t) (val[0] * (int16_t) (job.w1 << (6 - job.wbits)));
            int16_t t2 = (int16_t) (val[1] * (int16_t) (job.w2 << (6 - job.wbits)));
            t1 = (int16_t) (t1 + t2);
            t1 = (int16_t) (t1 + 32);
            pred = (int16_t) (t1 >> 6);
          } else {
            // orc_combine2_nxm_u8, schroorc.orc:1737-1757
            int16_t t1 = (int16_t) (val[0] * job.w1);
            int16_t t2 = (int16_t) (val[1] * job.w2);
            t1 = (int16_t) (t1 + t2);
            t1 = (int16_t) (t1 + (int16_t) ((1 << job.wbits) >> 1));
            t1 = (int16_t) (t1 >> job.wbits);
            pred = clampi (t1, 0, 255);
          }
        } else {
          int s = val[mode - 1];
          if constexpr (SIMPLE) {
            pred = s;
          } else if (interior) {
            // block_acc_scaled, schromotion8.c:44-73
            int16_t t1 = (int16_t) (s * (int16_t) (wsum << (6 - job.wbits)));
            t1 = (int16_t) (t1 + 32);
            pred = (int16_t) (t1 >> 6);
          } else if (noscale) {
            pred = s;
          } else {
            // get_ref1_block: ROUND_SHIFT stored into a uint8_t, schromotion8.c:391-397
            pred = (uint8_t) ((s * wsum + (1 << (job.wbits - 1))) >> job.wbits);
          }
        }
      }
      acc += pred * wgt;        // only the low 16 bits matter
    }
  }

  if (job.out_s16) {
    // orc_rrshift6_s16_ip_2d (schroorc.orc:676-682; schromotion8.c:896-899 with add = FALSE): subw 8160, shrsw 6
    const int16_t t = (int16_t) ((int16_t) ((int16_t) acc - 8160) >> 6);
    gstore < int16_t > ((int16_t *) (job.out + (size_t) py * job.out_stride) + px, t);
    return;
  }
  // orc_rrshift6_add_s16_2d / _s32_2d
  int16_t t1 = (int16_t) ((int16_t) acc + 32);
  t1 = (int16_t) (t1 >> 6);
  int16_t res;
  if (!job.residual)            // no residual to add (a zero_residual picture, schrodecoder.c:1904-1906)
    res = 0;
  else if (job.res_bpp == 2)
    res = gload < int16_t > ((const int16_t *) ((const char *) job.residual
            + (size_t) py * job.residual_stride) + px);
  else
    res = (int16_t) gload < int32_t > ((const int32_t *) ((const char *) job.residual
            + (size_t) py * job.residual_stride) + px);
  t1 = (int16_t) (res + t1);
  gstore < uint8_t > (job.out + (size_t) py * job.out_stride + px, (uint8_t) clampi (t1, 0, 255));
}

// ---------------------------------------------------------------------------
// Item kernel for the default picture weights (1,1,bits 1 -- "simple_weight",
// schromotion8.c:773-776), the case every stream in the reference's test
// suite uses.  With those weights edge and interior blocks predict the same
// value, so the only edge special-case left is the weight folding.
//
// One 256-thread workgroup owns a 128x32 output tile and an int accumulator
// tile in LDS.  Work items are (block, block row); a lane takes a 4-pixel
// segment of one: it fetches the segment's reference samples with two unaligned
// 8-byte loads per reference (quarter/eighth-pel: v_perm_b32 + v_dot4_u32_u8 per
// pixel), forms the prediction once, and adds pred*wx*wy into the LDS tile
// (ds_add_u32; all adds are modulo 2^16 in the reference, so order is
// irrelevant).  The MV decode and clamp are done once per block and tile.  The
// finish pass reads the residual (8-byte loads), rounds, adds, clamps and
// stores 4 pixels per lane.  (How the work list is built: see obmc_item_kernel.)

constexpr int kFTW = 128, kFTH = 32;


constexpr int kAccStride = 136; // 4 + 128 + 4; 8 mod 64: block rows spread over the LDS banks, rows 16-byte aligned

// One decoded block.  Everything that is uniform over the block's pixels is
// worked out once here: position, prediction mode, get_block's clamped fetch
// origin, the byte offset of its first reference sample, the packed bilinear
// weights, and whether the whole sample window lies inside the half-pel image
// (then no per-sample clamp is needed).
struct BlkInfo {
  int bx, by;
  int mode_dc;                  // bits 0-1 mode, bit 2/3: ref 0/1 window needs clamping, bits 8..: DC value
  int fx[2], fy[2];             // clamped (px, py) per reference, 1/2^prec pel
  int off[2];                   // byte offset of the first sample (window inside the image)
  uint32_t wpk[2];              // (w00, w01, w10, w11), orc_combine4_nxm_u8
  int pad;
};

// four horizontally adjacent samples from a window known to be inside the image
template < int PC >
__device__ __forceinline__ void
fetch4_inside (const uint8_t * __restrict__ p, int stride, uint32_t wpk, int *val)
{
  if constexpr (PC == 0) {
    uint32_t v = gload < u32_u > (p);
    val[0] = v & 0xff;
    val[1] = (v >> 8) & 0xff;
    val[2] = (v >> 16) & 0xff;
    val[3] = v >> 24;
  } else if constexpr (PC == 1) {
    const u32x2 q = gload < u32x2_u > (p);
    const uint64_t v = q.x | ((uint64_t) q.y << 32);
    val[0] = (int) (v & 0xff);
    val[1] = (int) ((v >> 16) & 0xff);
    val[2] = (int) ((v >> 32) & 0xff);
    val[3] = (int) ((v >> 48) & 0xff);
  } else {
    const u32x2 a = gload < u32x2_u > (p), b = gload < u32x2_u > (p + stride);
    const uint32_t alo = a.x, ahi = a.y, blo = b.x, bhi = b.y;
    val[0] = (int) (__builtin_amdgcn_udot4 (__builtin_amdgcn_perm (blo, alo, 0x05040100u), wpk, 8u, false) >> 4);
    val[1] = (int) (__builtin_amdgcn_udot4 (__builtin_amdgcn_perm (blo, alo, 0x07060302u), wpk, 8u, false) >> 4);
    val[2] = (int) (__builtin_amdgcn_udot4 (__builtin_amdgcn_perm (bhi, ahi, 0x05040100u), wpk, 8u, false) >> 4);
    val[3] = (int) (__builtin_amdgcn_udot4 (__builtin_amdgcn_perm (bhi, ahi, 0x07060302u), wpk, 8u, false) >> 4);
  }
}

constexpr int kAccMargin = 4;   // a 4-pixel segment may stick out of the tile by 3 pixels; 4 keeps rows aligned

struct TileCtx {
  int x_lo, x_hi, y_lo, y_hi;
  int xfold_hi, yfold_hi;
};

// The accumulator tile folds rows y and y + kFTH/2 into one LDS word (low / high half):
// the sums are 16-bit quantities (the reference accumulates in s16) and a sum of
// pred * weight with pred <= 255 and weights adding up to 64 never carries out of its
// half, so the hot path adds with one ds_add_u32.  This form is exact for ANY operand
// (a DC value outside 0..255 makes the reference's s16 sum wrap): it adds inside the
// chosen half only.
__device__ __forceinline__ void
acc_add_wrap (int *word, int high, int value)
{
  unsigned int old = __hip_atomic_load ((unsigned int *) word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), assumed;
  do {
    assumed = old;
    const unsigned int upd = high ? (assumed & 0xffffu) | ((assumed + ((unsigned int) value << 16)) & 0xffff0000u)
        : (assumed & 0xffff0000u) | ((assumed + (unsigned int) value) & 0xffffu);
    old = atomicCAS ((unsigned int *) word, assumed, upd);
  } while (old != assumed);
}

// Generic item: border blocks (per-sample clamp) and picture-edge weight
// folding (accumulate_slow, schromotion8.c:673-693).  Rare, not tuned.
template < int PC >
__device__ __forceinline__ void
obmc_item_slow (const ObmcJob & job, const BlkInfo & bi, int row, int seg, const TileCtx & tc,
    const int *s_wx, const int *s_wy, int *acc, bool exact)
{
  const int prec = job.prec;
  const int y = bi.by + row, xs = bi.bx + 4 * seg;
  const int md = bi.mode_dc, mode = md & 3;
  int pred[4];
  if (mode == 0) {
    pred[0] = pred[1] = pred[2] = pred[3] = md >> 8;
  } else {
    int val[2][4];
    for (int r = 0; r < 2; r++) {
      if (!(mode & (r + 1)))
        continue;
      if constexpr (PC == 2) {
        // the four pixels are 8 eighth-pels apart: they share the blend weights and read nine
        // adjacent half-pel columns of two rows (each clamped on its own, as fetch_ref does)
        const int sx = bi.fx[r] + 4 * seg * (1 << prec), sy = bi.fy[r] + row * (1 << prec);
        const int x8 = prec == 2 ? sx * 2 : sx, y8 = prec == 2 ? sy * 2 : sy;
        const int hx = x8 >> 2, hy = y8 >> 2, rx = x8 & 3, ry = y8 & 3;
        const int Y0 = clampi (hy, 0, 2 * job.h - 2), Y1 = clampi (hy + 1, 0, 2 * job.h - 2);
        int p[2][9];
#pragma unroll
        for (int j = 0; j < 9; j++) {
          const int X = clampi (hx + j, 0, 2 * job.w - 2);
          p[0][j] = gload < uint8_t > (job.ref[r] + hp_offset (X, Y0, job.ref_stride[r], job.ref_ps, job.ref_cb));
          p[1][j] = gload < uint8_t > (job.ref[r] + hp_offset (X, Y1, job.ref_stride[r], job.ref_ps, job.ref_cb));
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int v = (4 - ry) * ((4 - rx) * p[0][2 * e] + rx * p[0][2 * e + 1])
              + ry * ((4 - rx) * p[1][2 * e] + rx * p[1][2 * e + 1]);
          val[r][e] = (v + 8) >> 4;
        }
      } else {
        for (int e = 0; e < 4; e++)
          val[r][e] = fetch_ref < PC > (job.ref[r], job.ref_stride[r], job.w, job.h,
              bi.fx[r] + (4 * seg + e) * (1 << prec), bi.fy[r] + row * (1 << prec), prec, job.ref_ps, job.ref_cb);
      }
    }
    for (int e = 0; e < 4; e++)
      pred[e] = mode == 3 ? (val[0][e] + val[1][e] + 1) >> 1 : (mode == 1 ? val[0][e] : val[1][e]);
  }
  int wy = s_wy[row];
  if (y < job.yoff)
    wy += s_wy[2 * job.yoff - row - 1];
  if (y >= tc.yfold_hi)
    wy += s_wy[2 * (job.yblen - job.yoff) - row - 1];
  const int yrel = y - tc.y_lo;
  int *arow = acc + (yrel & (kFTH / 2 - 1)) * kAccStride + kAccMargin - tc.x_lo;
  for (int e = 0; e < 4; e++) {
    const int x = xs + e, idx = 4 * seg + e;
    if (idx >= job.xblen)
      continue;
    int wx = s_wx[idx];
    if (x < job.xoff)
      wx += s_wx[2 * job.xoff - idx - 1];
    if (x >= tc.xfold_hi)
      wx += s_wx[2 * (job.xblen - job.xoff) - idx - 1];
    // plain add into the row's half of the word unless some sum of the tile may wrap (see the kernel)
    if (exact)
      acc_add_wrap (arow + x, yrel & (kFTH / 2), pred[e] * wx * wy);
    else
      atomicAdd (arow + x, (pred[e] * wx * wy) << (yrel & (kFTH / 2)));
  }
}

// out = sat_u8 (residual + ((acc + 32) >> 6)) for one tile, 4 pixels per lane
__device__ __forceinline__ void
obmc_finish (const ObmcJob & job, const int *acc, int tid, int x_lo, int y_lo, int x_hi, int y_hi)
{
  typedef short s16x2 __attribute__ ((ext_vector_type (2)));
  // whole-width tile, s16 residual, aligned rows: 8 pixels per lane with packed 16-bit
  // arithmetic (the reference's adds wrap at 16 bits: v_pk_add_u16 does exactly that)
  if (job.out_s16) {
    // the prediction - 128 as s16 (orc_rrshift6_s16_ip_2d: subw 8160, shrsw 6), a pixel per lane and step
    for (int it = tid; it < kFTH * kFTW; it += kThreads) {
      const int xx = it % kFTW, yy = it / kFTW;
      const int x = x_lo + xx, y = y_lo + yy;
      if (y >= y_hi || x >= x_hi)
        continue;
      const int a = acc[(yy & (kFTH / 2 - 1)) * kAccStride + kAccMargin + xx] >> (yy & (kFTH / 2));
      gstore < int16_t > ((int16_t *) (job.out + (size_t) y * job.out_stride) + x, (int16_t) ((int16_t) ((int16_t) a - 8160) >> 6));
    }
    return;
  }
  const bool fast = job.res_bpp == 2 && x_hi - x_lo == kFTW
      && ((((uintptr_t) job.residual) | (uintptr_t) job.residual_stride) & 15) == 0
      && ((((uintptr_t) job.out) | (uintptr_t) job.out_stride) & 7) == 0;
  if (fast) {
    // one lane: 8 pixels of row yy (low halves) and of row yy + kFTH/2 (high halves)
    for (int it = tid; it < (kFTH / 2) * (kFTW / 8); it += kThreads) {
      const int g = it & (kFTW / 8 - 1), yy = it >> 4;
      static_assert (kFTW / 8 == 16, "8-pixel groups per tile row");
      const int x = x_lo + 8 * g;
      const int4 *ap = reinterpret_cast < const int4 * >(acc + yy * kAccStride + kAccMargin + 8 * g);
      const int4 a0 = ap[0], a1 = ap[1];
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const int y = y_lo + yy + half * (kFTH / 2);
        if (y >= y_hi)
          continue;
        const uint32_t sel = half ? 0x07060302u : 0x05040100u;
        const u32x4 r = job.residual ? gload < u32x4 > ((const char *) job.residual + (size_t) y * job.residual_stride + 2 * x)
            : (u32x4) { 0u, 0u, 0u, 0u };
        const uint32_t av[4] = {
          __builtin_amdgcn_perm ((uint32_t) a0.y, (uint32_t) a0.x, sel),
          __builtin_amdgcn_perm ((uint32_t) a0.w, (uint32_t) a0.z, sel),
          __builtin_amdgcn_perm ((uint32_t) a1.y, (uint32_t) a1.x, sel),
          __builtin_amdgcn_perm ((uint32_t) a1.w, (uint32_t) a1.z, sel)
        };
        const uint32_t rv[4] = { r.x, r.y, r.z, r.w };
        uint32_t t[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          s16x2 v = (__builtin_bit_cast (s16x2, av[k]) + (short) 32) >> 6;
          v = v + __builtin_bit_cast (s16x2, rv[k]);
          v = __builtin_elementwise_min (__builtin_elementwise_max (v, (s16x2) (short) 0), (s16x2) (short) 255);
          t[k] = __builtin_bit_cast (uint32_t, v);
        }
        u32x2 o;
        o.x = __builtin_amdgcn_perm (t[1], t[0], 0x06040200u);
        o.y = __builtin_amdgcn_perm (t[3], t[2], 0x06040200u);
        gstore < u32x2 > (job.out + (size_t) y * job.out_stride + x, o);
      }
    }
    return;
  }
  // orc_rrshift6_add_s16_2d / _s32_2d on 4 pixels per lane
  for (int it = tid; it < kFTH * kFTW / 4; it += kThreads) {
    const int g = it % (kFTW / 4), yy = it / (kFTW / 4);
    const int x = x_lo + 4 * g, y = y_lo + yy;
    if (y >= y_hi || x >= x_hi)
      continue;
    const int *ap = acc + (yy & (kFTH / 2 - 1)) * kAccStride + kAccMargin + 4 * g;
    const int hs = yy & (kFTH / 2);     // 0 or 16: which half of the word
    const int av[4] = { ap[0] >> hs, ap[1] >> hs, ap[2] >> hs, ap[3] >> hs };
    const char *rrow = (const char *) job.residual + (size_t) y * job.residual_stride;
    uint8_t *orow = job.out + (size_t) y * job.out_stride + x;
    __attribute__ ((aligned (8))) int16_t res[4];
    const bool full = x + 4 <= job.w;
    if (!job.residual) {
      res[0] = res[1] = res[2] = res[3] = 0;
    } else if (job.res_bpp == 2) {
      const int16_t *rp = (const int16_t *) rrow + x;
      if (full && (((uintptr_t) rp) & 7) == 0) {
        *reinterpret_cast < u32x2 * >(res) = gload < u32x2 > (rp);
      } else {
        for (int e = 0; e < 4; e++)
          res[e] = x + e < job.w ? gload < int16_t > (rp + e) : (int16_t) 0;
      }
    } else {
      const int32_t *rp = (const int32_t *) rrow + x;
      for (int e = 0; e < 4; e++)
        res[e] = x + e < job.w ? (int16_t) gload < int32_t > (rp + e) : (int16_t) 0;   // convlw
    }
    uint32_t pk = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      int16_t t1 = (int16_t) ((int16_t) av[e] + 32);
      t1 = (int16_t) (t1 >> 6);
      t1 = (int16_t) (res[e] + t1);
      pk |= (uint32_t) clampi (t1, 0, 255) << (8 * e);
    }
    if (full && (((uintptr_t) orow) & 3) == 0) {
      gstore < uint32_t > (orow, pk);
    } else {
      for (int e = 0; e < 4 && x + e < job.w; e++)
        gstore < uint8_t > (orow + e, (uint8_t) (pk >> (8 * e)));
    }
  }
}

// ---------------------------------------------------------------------------
// The work list.  Blocks are decoded once per tile (one per thread), sorted by class
// (two references / first / second / DC / needs the exact rim path), and a prefix sum over
// "rows of the block inside the tile" turns them into a dense list of ITEMS = (block, row).
// nseg = ceil (xblen / 4) adjacent lanes take one item, 4 pixels each, so every lane of a
// pass has work and all lanes of a pass belong to one class: the pass body is straight
// line (loads of both references issued together, one wait, v_perm/v_dot4 bilinear,
// v_mul_u32_u24 with the 4 packed weights of the (row, segment), 4 ds_add_u32).
// Everything that does not depend on the pixel is precomputed per block (tile-relative
// origin, sample-window offsets, packed bilinear weights) or per (row, segment) (weights).
//
// This form replaced a kernel whose lanes were fixed (block slot, row, segment) triples:
// 44 % of its lanes were rows outside the tile and every item walked a chain of divergent
// range / mode / clamp tests (3050 wave instructions per wave, a third of them scalar
// exec-mask bookkeeping; this one 2000).  Both run 8 x 2160p in 0.55 ms: the launch is
// bound by the cache-line gather itself -- 47 M 128-byte lines per launch from L2 /
// Infinity Cache for 24 useful bytes each, 11 TB/s of line traffic (rocprofv3
// TCP_TCC_READ_REQ; MI355X_MICROARCH.md gives 8.6 - 18 TB/s for gathers) -- not by
// instruction issue, occupancy (3, 4 or 5 workgroups per CU: same time) or LDS atomics.

constexpr int kItemBlkCap = 192;        // decoded blocks per chunk (<= kThreads: one per thread)
constexpr int kItemCap = 1536;          // (block, row) items per chunk
constexpr int kItemWCap = 256;          // (row, segment) weight words; larger blocks take the rim path

struct __attribute__ ((aligned (16))) HotBlk {
  int y, x;                     // block origin relative to the tile
  int mode_dc;                  // as BlkInfo
  int rows;                     // first block row inside the tile | rows inside << 8 | window phases << 16
  int off[2];                   // first sample of the window: byte offset (plain plane); half-pel image: padded column of
                                // its plane | parities of the half-pel origin << 16 (x) / << 17 (y)
  uint32_t wpk[2];              // packed bilinear weights
  int ya[2];                    // half-pel image: plane row of the window's first sample
};

struct ItemLane {
  int slot, seg;                // item slot within the pass, 4-pixel segment within the row
  int seg_bytes;                // byte offset of the segment inside the sample window
  int tw3;                      // tile width + 3 (range test of a segment)
  bool active;                  // lanes beyond the last whole item of a pass idle
};

// Four predicted pixels of one reference from the half-pel planes (schro_hip_internal.h): the
// segment's samples are four contiguous bytes of the plane the window's origin selects, the other
// taps of a quarter / eighth-pel position four contiguous bytes of the neighbouring planes; one
// byte-aligned dword load each, then v_perm_b32 + v_dot4_u32_u8 per pixel (orc_combine4_nxm_u8).
template < int PC >
__device__ __forceinline__ void
hp_predict4 (const ObmcJob & job, int r, int off_r, int ya, int row, int seg, uint32_t wpk, int *val)
{
  static_assert (PC >= 1, "plain references are linear");
  const int stride = job.ref_stride[r];
  // pair images (ref_ps 1): a sample is two bytes, this component byte ref_cb of it -- the segment is
  // eight contiguous bytes (still inside its chunk: it starts at an even byte <= 14 and the X + 1 tap
  // two bytes on), of which every second one is taken
  const int ps = job.ref_ps;
  const int xp = ((off_r & 0xffff) + 4 * seg) << ps, px = (off_r >> 16) & 1, py = (off_r >> 17) & 1;
  const int y = ya + row;
  const uint32_t pick = job.ref_cb ? 0x07050301u : 0x06040200u;
  auto load4 = [&](const uint8_t * p) {
    if (ps) {
      const u32x2 q = gload < u32x2_u > (p);
      return __builtin_amdgcn_perm (q.y, q.x, pick);
    }
    return gload < u32_u > (p);
  };
  const uint8_t *a = job.ref[r] + hp_row_offset (y, stride) + hp_col_offset (xp) + (px + 2 * py) * 128;
  const uint32_t A = load4 (a);
  if constexpr (PC == 1) {
    val[0] = A & 0xff;
    val[1] = (A >> 8) & 0xff;
    val[2] = (A >> 16) & 0xff;
    val[3] = A >> 24;
    (void) wpk;
  } else {
    // X + 1: the other column parity, one sample on when X is odd (that sample is in the same
    // chunk: chunks hold 32 bytes and advance by 16); Y + 1: the other row parity, one row on when Y is odd
    const int dB = px ? (1 << ps) - 128 : 128;
    const uint8_t *c = job.ref[r] + hp_row_offset (y + py, stride) + hp_col_offset (xp) + (px + 2 * (py ^ 1)) * 128;
    const uint32_t B = load4 (a + dB), C = load4 (c), D = load4 (c + dB);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint32_t sel = 0x0c0c0000u | (uint32_t) e | ((uint32_t) (4 + e) << 8);
      const uint32_t t = __builtin_amdgcn_perm (B, A, sel), u = __builtin_amdgcn_perm (D, C, sel);
      val[e] = (int) (__builtin_amdgcn_udot4 (__builtin_amdgcn_perm (u, t, 0x05040100u), wpk, 8u, false) >> 4);
    }
  }
}

// one pass of one class: CLS 0 both references, 1 / 2 one reference, 3 DC
template < int PC, int CLS, bool EXACT >
__device__ __forceinline__ void
item_pass (const ObmcJob & job, const ItemLane & il, const uint16_t * s_item, const HotBlk * s_hot,
    const uint32_t * s_w4, int nseg, int *acc, int i0, int hi)
{
  constexpr int kStep = PC == 0 ? 1 : 2;
  const int it = i0 + il.slot;
  const int e = s_item[min (it, hi - 1)];
  const HotBlk & hb = s_hot[e & 0xff];
  const int row = e >> 8;
  const int yrel = hb.y + row, xrel = hb.x + 4 * il.seg;
  const uint32_t w4 = s_w4[__umul24 (row, nseg) + il.seg];
  int v0[4], v1[4];
  if constexpr (CLS == 3) {
    v0[0] = v0[1] = v0[2] = v0[3] = v1[0] = v1[1] = v1[2] = v1[3] = hb.mode_dc >> 8;
  } else {
    constexpr int r0 = CLS == 2 ? 1 : 0;
    if constexpr (PC == 0) {
      fetch4_inside < PC > (job.ref[r0] + (hb.off[r0] + (row * kStep) * job.ref_stride[r0] + il.seg_bytes),
          job.ref_stride[r0], hb.wpk[r0], v0);
      if constexpr (CLS == 0)
        fetch4_inside < PC > (job.ref[1] + (hb.off[1] + (row * kStep) * job.ref_stride[1] + il.seg_bytes),
            job.ref_stride[1], hb.wpk[1], v1);
    } else {
      hp_predict4 < PC > (job, r0, hb.off[r0], hb.ya[r0], row, il.seg, hb.wpk[r0], v0);
      if constexpr (CLS == 0)
        hp_predict4 < PC > (job, 1, hb.off[1], hb.ya[1], row, il.seg, hb.wpk[1], v1);
    }
    if constexpr (CLS != 0) {
#pragma unroll
      for (int k = 0; k < 4; k++)
        v1[k] = v0[k];
    }
  }
  if (il.active && it < hi && (unsigned) (xrel + 3) < (unsigned) il.tw3) {
    // rows yrel and yrel + 16 share a word (low / high half); pred <= 255 here
    int *ap = acc + (__umul24 (yrel & (kFTH / 2 - 1), kAccStride) + kAccMargin + xrel);
    const int hs = yrel & (kFTH / 2);
    static_assert (kFTH / 2 == 16, "the high half of an accumulator word is bit 16 up");
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int pred = CLS == 0 ? (v0[k] + v1[k] + 1) >> 1 : v0[k];
      const int v = __mul24 (pred, (int) ((w4 >> (8 * k)) & 0xff));
      if constexpr (EXACT)      // some sum of this tile may wrap: no plain adds (see the kernel)
        acc_add_wrap (ap + k, hs, v);
      else
        atomicAdd (ap + k, v << hs);
    }
  }
}

template < int PC, int CLS >
__device__ __forceinline__ void
item_class (const ObmcJob & job, const ItemLane & il, const uint16_t * s_item, const HotBlk * s_hot,
    const uint32_t * s_w4, int nseg, int *acc, int first, int hi, int stride, bool exact)
{
  if (exact) {                  // rare (see the kernel): keep it out of the hot loop
    for (int i0 = first; i0 < hi; i0 += stride)
      item_pass < PC, CLS, true > (job, il, s_item, s_hot, s_w4, nseg, acc, i0, hi);
  } else {
    for (int i0 = first; i0 < hi; i0 += stride)
      item_pass < PC, CLS, false > (job, il, s_item, s_hot, s_w4, nseg, acc, i0, hi);
  }
}

template < int PC >
__global__ __launch_bounds__ (kThreads) __attribute__ ((amdgpu_waves_per_eu (5, 5)))
void obmc_item_kernel (const ObmcJob * __restrict__ jobs, int njobs, const uint32_t * __restrict__ order,
    uint32_t * __restrict__ overflow)
{
  __shared__ __attribute__ ((aligned (16))) int acc[(kFTH / 2) * kAccStride];   // rows y and y + 16 per word
  __shared__ int s_wx[kMaxBlk], s_wy[kMaxBlk];
  __shared__ HotBlk s_hot[kItemBlkCap];
  __shared__ uint16_t s_item[kItemCap];
  __shared__ uint32_t s_w4[kItemWCap];
  __shared__ uint16_t s_start[kThreads + 2];    // first item of each sorted block
  __shared__ int s_cnt[8];
  __shared__ int s_wide;

  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  // which tile: position bid of the host's tile order (obmc_tile_order: the same rows of all
  // pictures that share references side by side), or of the plain plane-by-plane list
  const uint32_t entry = order ? __builtin_amdgcn_readfirstlane (gload < uint32_t > (order + bid)) : 0u;
  const ObmcJob job = jobs[order ? (int) (entry >> 16) : find_job (jobs, njobs, bid)];
  const int t = order ? (int) (entry & 0xffffu) : bid - job.tile_base;
  const int ty = mdiv (t, job.tiles_x, job.m_tiles_x), tx = t - ty * job.tiles_x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x_lo = tx * kFTW, y_lo = ty * kFTH;
  const int x_hi = min (x_lo + kFTW, job.w), y_hi = min (y_lo + kFTH, job.h);

  static_assert (((kFTH / 2) * kAccStride) % 4 == 0, "accumulator tile is cleared 16 bytes at a time");
  for (int it = tid; it < (kFTH / 2) * kAccStride / 4; it += kThreads)
    reinterpret_cast < int4 * >(acc)[it] = make_int4 (0, 0, 0, 0);
  if (tid < job.xblen)
    s_wx[tid] = obmc_weight_1d (tid, job.xblen, job.xoff);
  if (tid >= 64 && tid - 64 < job.yblen)
    s_wy[tid - 64] = obmc_weight_1d (tid - 64, job.yblen, job.yoff);

  const int xblen = job.xblen, yblen = job.yblen, xbsep = job.xbsep, ybsep = job.ybsep;
  const int xoff = job.xoff, yoff = job.yoff, prec = job.prec;
  // first / last block whose footprint meets the tile (numerators kept non-negative)
  const int i_lo = max (0, mdiv (x_lo + xoff - xblen + 2 * xbsep, xbsep, job.m_xbsep) - 1);
  const int i_hi = min (job.nbx - 1, mdiv (x_hi - 1 + xoff, xbsep, job.m_xbsep));
  const int j_lo = max (0, mdiv (y_lo + yoff - yblen + 2 * ybsep, ybsep, job.m_ybsep) - 1);
  const int j_hi = min (job.nby - 1, mdiv (y_hi - 1 + yoff, ybsep, job.m_ybsep));
  const int nbi = i_hi - i_lo + 1, nbj = j_hi - j_lo + 1;
  const int nblk = nbi > 0 && nbj > 0 ? nbi * nbj : 0;
  // blk / nbi: one table read per workgroup instead of a division per thread.  (r04: the 32-bit magic, exact for every block
  // count a tile can have.  r01's 16-bit form -- (blk * ceil (2^16 / nbi)) >> 16 -- is exact only while blk * nbi < 2^16: a
  // 128 x 32 tile of 4 x 4 blocks every 2 pixels meets 62 x 18 = 1116 of them, and the last one, blk 1115, came out a row
  // down and a column left of the grid; found by tests/test_gpu_fuzz.py, draw 4430 of seed 77 x 100.)
  const uint32_t m_nbi = nbi > 1 && nbi <= 1024 ? kDivMagic.m[nbi] : 0u;
  const int xfold_hi = job.nbx * xbsep - xoff, yfold_hi = job.nby * ybsep - yoff;
  const int gw = PC == 0 ? job.w - 1 : 2 * job.w - 2;   // last valid sample column
  const int gh = PC == 0 ? job.h - 1 : 2 * job.h - 2;
  constexpr int kStep = PC == 0 ? 1 : 2;        // samples per pixel step
  // block geometry of the plane: from the host (obmc_item_geometry)
  const int nseg = job.nseg, lpi = job.lpi;
  const int IPW = job.ipw, chunk_cap = job.chunk_cap;
  ItemLane il;
  il.slot = mdiv (lane, lpi, job.m_lpi);
  const int sub = lane - il.slot * lpi;
  il.seg = min (sub, nseg - 1);
  il.seg_bytes = il.seg * (4 * kStep);
  il.tw3 = x_hi - x_lo + 3;
  il.active = il.slot < IPW && sub < nseg;
  TileCtx tc;
  tc.x_lo = x_lo;
  tc.x_hi = x_hi;
  tc.y_lo = y_lo;
  tc.y_hi = y_hi;
  tc.xfold_hi = xfold_hi;
  tc.yfold_hi = yfold_hi;
  __syncthreads ();             // weights visible
  // the 4 weights wx * wy of every (block row, segment), one byte each (<= 64)
  for (int i = tid; i < yblen * nseg && i < kItemWCap; i += kThreads) {
    const int r = mdiv (i, nseg, job.m_nseg), sg = i - r * nseg;
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (4 * sg + k < xblen)
        w |= (uint32_t) (s_wx[4 * sg + k] * s_wy[r]) << (8 * k);
    s_w4[i] = w;
  }

  // The hot path adds into one half of an accumulator word with a plain ds_add_u32.  That is
  // exact as long as no 16-bit sum of the tile wraps, which only a DC value outside 0..255
  // can cause (never in a legal stream).  If a block has one, every add of the tile goes
  // through acc_add_wrap instead; found in a later chunk, the tile starts over that way.
  bool exact = false;
  for (int chunk0 = 0; chunk0 < nblk; chunk0 += chunk_cap) {
    const int nb = min (chunk_cap, nblk - chunk0);
    if (tid < 8)
      s_cnt[tid] = 0;
    if (tid == 8)
      s_wide = 0;
    __syncthreads ();           // acc / weights ready; previous chunk's tables consumed
    // ---- decode one block per thread; class 0 two references, 1 / 2 one reference,
    // 3 DC, 4 picture rim (exact clamp / fold path) -------------------------------
    HotBlk info;
    int key = 0, rank = 0;
    const bool have = tid < nb;
    if (have) {
      const int blk = chunk0 + tid;
      const int bj = nbi == 1 ? blk : nbi <= 1024 ? (int) __umulhi ((uint32_t) blk, m_nbi) : blk / nbi;
      const int i = i_lo + (blk - bj * nbi), jj = j_lo + bj;
      const uint8_t *mvp = job.mvs + (size_t) 20 * ((size_t) jj * job.nbx + i);
      const uint32_t flags = gload < uint32_t > (mvp);
      const uint32_t v01 = gload < uint32_t > (mvp + 12);
      const uint32_t v23 = gload < uint32_t > (mvp + 16);
      const int bx = xbsep * i - xoff, by = ybsep * jj - yoff;
      info.x = bx - x_lo;
      info.y = by - y_lo;
      const int mode = flags & 3;
      const bool interior = i >= 1 && i < job.max_x_blocks && jj >= 1 && jj < job.max_y_blocks;
      int dc = job.comp == 0 ? (int16_t) (v01 & 0xffff)
          : job.comp == 1 ? (int16_t) (v01 >> 16) : (int16_t) (v23 & 0xffff);
      // get_dc_block stores a uint8_t; block_acc_dc multiplies a 16-bit parameter
      int p = interior ? (int) (int16_t) (dc + 128) : (int) (uint8_t) (dc + 128);
      int md = mode | (p << 8);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        int fx, fy;
        mv_origin (job, bx, by, v01, v23, r, &fx, &fy);
        int gx0, gy0, gx1, gy1;
        uint32_t wpk = 0;
        if constexpr (PC == 2) {
          const int x8 = prec == 2 ? fx * 2 : fx, y8 = prec == 2 ? fy * 2 : fy;
          const int rx = x8 & 3, ry = y8 & 3;
          gx0 = x8 >> 2;
          gy0 = y8 >> 2;
          gx1 = gx0 + 2 * (nseg * 4 - 1) + 1;   // whole 8-byte loads of the last segment
          gy1 = gy0 + 2 * (yblen - 1) + 1;
          wpk = (uint32_t) ((4 - ry) * (4 - rx)) | ((uint32_t) ((4 - ry) * rx) << 8)
              | ((uint32_t) (ry * (4 - rx)) << 16) | ((uint32_t) (ry * rx) << 24);
        } else {
          gx0 = fx;
          gy0 = fy;
          gx1 = gx0 + kStep * (nseg * 4 - 1);
          gy1 = gy0 + kStep * (yblen - 1);
        }
        bool inside = gx0 >= 0 && gy0 >= 0 && gx1 <= gw && gy1 <= gh;
        if (!inside)
          md |= 4 << r;
        // plain planes are linear; half-pel planes: padded column and row of the first sample in its plane
        if constexpr (PC == 0) {
          info.off[r] = inside ? gy0 * job.ref_stride[r] + gx0 : 0;
          info.ya[r] = 0;
        } else {
          info.off[r] = inside ? ((gx0 >> 1) + kHpApron) | ((gx0 & 1) << 16) | ((gy0 & 1) << 17) : 0;
          info.ya[r] = inside ? gy0 >> 1 : 0;
        }
        info.wpk[r] = wpk;
      }
      info.mode_dc = md;
      const int ra = max (0, -info.y), rb = min (yblen, y_hi - by);
      info.rows = ra | ((rb - ra) << 8);
      const bool clamped = ((md >> 2) & md & 3) != 0;
      const bool fold = by < yoff || by + yblen > yfold_hi || bx < xoff || bx + nseg * 4 > xfold_hi;
      // (a DC value outside 0..255 would carry between the halves of an accumulator word)
      const bool wide_dc = mode == 0 && (unsigned) p > 255u;
      if (wide_dc) {
        s_wide = 1;
        if (overflow)
          __hip_atomic_store (overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      key = (clamped || fold || wide_dc || yblen * nseg > kItemWCap) ? 4 : (mode == 3 ? 0 : (mode == 0 ? 3 : mode));
      if (key == 4) {
        // the rim path works from the clamped fetch origins (it has no use for the window
        // offsets and blend weights): keep them, so that it need not read the vectors again
        int fx, fy;
        mv_origin (job, bx, by, v01, v23, 0, &fx, &fy);
        info.off[0] = fx;
        info.off[1] = fy;
        mv_origin (job, bx, by, v01, v23, 1, &fx, &fy);
        info.wpk[0] = (uint32_t) fx;
        info.wpk[1] = (uint32_t) fy;
      }
      rank = atomicAdd (&s_cnt[key], 1);
    }
    __syncthreads ();
    if (s_wide && !exact) {
      exact = true;
      if (chunk0 > 0) {         // adds already made may have carried: start the tile over
        __syncthreads ();
        for (int it = tid; it < (kFTH / 2) * kAccStride / 4; it += kThreads)
          reinterpret_cast < int4 * >(acc)[it] = make_int4 (0, 0, 0, 0);
        chunk0 = -chunk_cap;
        continue;
      }
    }
    int cbase[6];               // first sorted position of each class
    cbase[0] = 0;
#pragma unroll
    for (int k = 0; k < 5; k++)
      cbase[k + 1] = cbase[k] + s_cnt[k];
    if (have) {
      int base = 0;
#pragma unroll
      for (int k = 0; k < 5; k++)
        base = key == k ? cbase[k] : base;
      s_hot[base + rank] = info;
    }
    __syncthreads ();
    // ---- items: exclusive prefix sum of rows-in-tile over the sorted fast blocks --------
    int nrows = tid < cbase[4] ? ((s_hot[tid].rows >> 8) & 0xff) : 0, incl = nrows;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up (incl, d);
      if (lane >= d)
        incl += up;
    }
    if (lane == 63)
      s_cnt[4 + wave] = incl;   // s_cnt[4] (the rim class count) is already in cbase
    __syncthreads ();
    {
#pragma unroll
      for (int w = 0; w < kThreads / 64 - 1; w++)
        incl += wave > w ? s_cnt[4 + w] : 0;
      const int start = incl - nrows;
      s_start[tid] = (uint16_t) start;
      if (tid == kThreads - 1)
        s_start[kThreads] = (uint16_t) incl;
      if (tid < cbase[4]) {
        const int ra = s_hot[tid].rows & 0xff;
        for (int r = 0; r < nrows; r++)
          s_item[start + r] = (uint16_t) (tid | ((ra + r) << 8));
      }
    }
    __syncthreads ();

    // ---- accumulate: every pass of a wave is IPW whole items of one class ---------
    const int first = wave * IPW, stride = (kThreads / 64) * IPW;
    item_class < PC, 0 > (job, il, s_item, s_hot, s_w4, nseg, acc, s_start[cbase[0]] + first, s_start[cbase[1]], stride, exact);
    item_class < PC, 1 > (job, il, s_item, s_hot, s_w4, nseg, acc, s_start[cbase[1]] + first, s_start[cbase[2]], stride, exact);
    item_class < PC, 2 > (job, il, s_item, s_hot, s_w4, nseg, acc, s_start[cbase[2]] + first, s_start[cbase[3]], stride, exact);
    item_class < PC, 3 > (job, il, s_item, s_hot, s_w4, nseg, acc, s_start[cbase[3]] + first, s_start[cbase[4]], stride, exact);

    // ---- picture-rim blocks: exact clamp / fold path -----------------------------
    {
      const int per_block = yblen * nseg;
      const uint32_t m_per_block = div_magic (per_block);     // (uniform: once per workgroup)
      const int nslow = cbase[5] - cbase[4];
      for (int item = tid; item < nslow * per_block; item += kThreads) {
        const int b = mdiv (item, per_block, m_per_block);
        const int rem = item - b * per_block;
        const int r2 = mdiv (rem, nseg, job.m_nseg), s2 = rem - r2 * nseg;
        const HotBlk & hb = s_hot[cbase[4] + b];
        BlkInfo bi;
        bi.bx = hb.x + x_lo;
        bi.by = hb.y + y_lo;
        const int y = bi.by + r2, xs = bi.bx + 4 * s2;
        if (y < y_lo || y >= y_hi || xs + 3 < x_lo || xs >= x_hi)
          continue;
        bi.mode_dc = hb.mode_dc;
        bi.fx[0] = hb.off[0];   // (kept by the decode for rim blocks)
        bi.fy[0] = hb.off[1];
        bi.fx[1] = (int) hb.wpk[0];
        bi.fy[1] = (int) hb.wpk[1];
        obmc_item_slow < PC > (job, bi, r2, s2, tc, s_wx, s_wy, acc, exact);
      }
    }
  }
  __syncthreads ();
  obmc_finish (job, acc, tid, x_lo, y_lo, x_hi, y_hi);
}

template < int PC >
int
launch_one (hipStream_t stream, const ObmcJob * d_jobs, int njobs, int total_tiles, int variant,
    const uint32_t * d_order, uint32_t * overflow)
{
  // SCHRO_HIP_OBMC_LDS_PAD (bytes of unused dynamic LDS): fewer workgroups per CU than the
  // five that fit, to leave registers for a kernel on the other queue (experiments)
  static const int lds_pad = SCHRO_ENV ("SCHRO_HIP_OBMC_LDS_PAD") ? atoi (SCHRO_ENV ("SCHRO_HIP_OBMC_LDS_PAD")) : 0;
  if (variant == 1)
    SCHRO_LAUNCH ((obmc_item_kernel < PC >), dim3 (total_tiles), dim3 (kThreads), lds_pad, stream,
        d_jobs, njobs, d_order, overflow);
  else
    SCHRO_LAUNCH ((obmc_kernel < PC, false >), dim3 (total_tiles), dim3 (kThreads), 0,
        stream, d_jobs, njobs, overflow);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "obmc launch: %s", hipGetErrorString (e));
  return 0;
}

}                               // namespace

// geometry of the item kernel for one plane (the kernel reads it from the job)
void
obmc_item_geometry (ObmcJob * j)
{
  const int nseg = (j->xblen + 3) >> 2;
  const int lpi = nseg;         // an item's lanes: one per 4-pixel segment
  j->nseg = nseg;
  j->lpi = lpi;
  j->ipw = 64 / lpi;            // items per wave pass
  j->chunk_cap = std::min (kItemBlkCap, kItemCap / std::min (j->yblen, kFTH));
  j->m_tiles_x = div_magic (j->tiles_x);
  j->m_xbsep = div_magic (j->xbsep);
  j->m_ybsep = div_magic (j->ybsep);
  j->m_nseg = div_magic (nseg);
  j->m_lpi = div_magic (lpi);
  j->m_xramp = div_magic (2 * j->xoff - 1);
  j->m_yramp = div_magic (2 * j->yoff - 1);
}

// variant 0: per-pixel kernel (any weights), 64x4 tiles
// variant 1: LDS-accumulate item kernel (default weights), 128x32 tiles
// variant 3 / 4: row kernels (obmc_row.hip): 128x32; (U, V) pairs from pair images 64x32
void
obmc_tiles (int variant, int w, int h, int xoff, int *tiles_x, int *tiles_y)
{
  (void) xoff;
  if (variant >= 1) {
    const int tw = variant >= 3 ? obmc_row_tile_width (variant == 4) : kFTW;
    const int th = variant >= 3 ? obmc_row_tile_height () : kFTH;
    *tiles_x = (w + tw - 1) / tw;
    *tiles_y = (h + th - 1) / th;
  } else {
    *tiles_x = (w + kTW - 1) / kTW;
    *tiles_y = (h + kTH - 1) / kTH;
  }
}

int
launch_obmc (hipStream_t stream, const ObmcJob * d_jobs, int njobs, int total_tiles, int prec,
    int variant, const uint32_t * d_order, uint32_t * overflow)
{
  switch (prec == 0 ? 0 : (prec == 1 ? 1 : 2)) {
    case 0: return launch_one < 0 > (stream, d_jobs, njobs, total_tiles, variant, d_order, overflow);
    case 1: return launch_one < 1 > (stream, d_jobs, njobs, total_tiles, variant, d_order, overflow);
    default: return launch_one < 2 > (stream, d_jobs, njobs, total_tiles, variant, d_order, overflow);
  }
}

}                               // namespace schro
