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

namespace schro {
namespace {

constexpr int kThreads = 256;
constexpr int kTW = 64, kTH = 4;
constexpr int kMaxBlk = 64;     // SCHRO_LIMIT_BLOCK_SIZE, schrolimits.h:67

__device__ __forceinline__ int
clampi (int x, int lo, int hi)
{
  return min (max (x, lo), hi);
}

// schromotion.c:40-49
__device__ int
get_ramp (int x, int offset)
{
  if (offset == 1)
    return x == 0 ? 3 : 5;
  return 1 + (6 * x + offset - 1) / (2 * offset - 1);
}

// schromotion.c:57-69
__device__ int
obmc_weight_1d (int i, int blen, int offset)
{
  if (offset == 0)
    return 8;
  if (i < 2 * offset)
    return get_ramp (i, offset);
  if (blen - 1 - i < 2 * offset)
    return get_ramp (blen - 1 - i, offset);
  return 8;
}

// One reference sample at (sx, sy) in 1/2^prec pel units.
// PC 0: plain plane.  PC 1: half-pel image.  PC 2: 1/4- or 1/8-pel bilinear
// of four half-pel samples (orc_combine4_nxm_u8, schroorc.orc:1635-1662; the
// avg2 / copy special cases of schroframe.c:2306-2350 are the same formula).
template < int PC >
__device__ __forceinline__ int
fetch_ref (const uint8_t * __restrict__ ref, int stride, int w, int h, int sx, int sy, int prec)
{
  if constexpr (PC == 0) {
    int X = clampi (sx, 0, w - 1), Y = clampi (sy, 0, h - 1);
    return gload < uint8_t > (ref + (size_t) Y * stride + X);
  } else if constexpr (PC == 1) {
    int X = clampi (sx, 0, 2 * w - 2), Y = clampi (sy, 0, 2 * h - 2);
    return gload < uint8_t > (ref + (size_t) Y * stride + X);
  } else {
    int x8 = prec == 2 ? sx * 2 : sx, y8 = prec == 2 ? sy * 2 : sy;
    int hx = x8 >> 2, hy = y8 >> 2, rx = x8 & 3, ry = y8 & 3;
    int X0 = clampi (hx, 0, 2 * w - 2), X1 = clampi (hx + 1, 0, 2 * w - 2);
    int Y0 = clampi (hy, 0, 2 * h - 2), Y1 = clampi (hy + 1, 0, 2 * h - 2);
    const uint8_t *r0 = ref + (size_t) Y0 * stride, *r1 = ref + (size_t) Y1 * stride;
    int p00 = gload < uint8_t > (r0 + X0), p01 = gload < uint8_t > (r0 + X1);
    int p10 = gload < uint8_t > (r1 + X0), p11 = gload < uint8_t > (r1 + X1);
    int v = (4 - ry) * ((4 - rx) * p00 + rx * p01) + ry * ((4 - rx) * p10 + rx * p11);
    return (v + 8) >> 4;
  }
}

template < int PC, bool SIMPLE >
__global__ __launch_bounds__ (kThreads)
void obmc_kernel (const ObmcJob * __restrict__ jobs, int njobs)
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
          val[r] = fetch_ref < PC > (job.ref[r], job.ref_stride[r], job.w, job.h, sx, sy, prec);
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

  // orc_rrshift6_add_s16_2d / _s32_2d
  int16_t t1 = (int16_t) ((int16_t) acc + 32);
  t1 = (int16_t) (t1 >> 6);
  int16_t res;
  if (job.res_bpp == 2)
    res = gload < int16_t > ((const int16_t *) ((const char *) job.residual
            + (size_t) py * job.residual_stride) + px);
  else
    res = (int16_t) gload < int32_t > ((const int32_t *) ((const char *) job.residual
            + (size_t) py * job.residual_stride) + px);
  t1 = (int16_t) (res + t1);
  gstore < uint8_t > (job.out + (size_t) py * job.out_stride + px, (uint8_t) clampi (t1, 0, 255));
}

// ---------------------------------------------------------------------------
// Tile kernel for the default picture weights (1,1,bits 1 -- "simple_weight",
// schromotion8.c:773-776), the case every stream in the reference's test
// suite uses.  With those weights edge and interior blocks predict the same
// value, so the only edge special-case left is the weight folding.
//
// One 256-thread workgroup owns a 64x32 output tile and an int accumulator
// tile in LDS.  Work items are (block, block row, 4-pixel segment): each lane
// fetches its segment's reference samples with two unaligned 8-byte loads per
// reference (quarter/eighth-pel: v_perm_b32 + v_dot4_u32_u8 per pixel), forms
// the prediction once, and adds pred*wx*wy into the LDS tile (ds_add_u32; all
// adds are modulo 2^16 in the reference, so order is irrelevant).  The MV
// decode and clamp are amortised over 4 pixels and every block row is touched
// once per tile instead of once per pixel.  The finish pass reads the residual
// (8-byte loads), rounds, adds, clamps and stores 4 pixels per lane.

constexpr int kFTW = 128, kFTH = 32;


__device__ __forceinline__ int
floor_div (int a, int b)
{
  int q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}

constexpr int kAccStride = 141; // odd: block rows land on different LDS banks
constexpr int kBlkCap = 256;    // decoded blocks held in LDS per chunk

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

// border blocks: per-sample coordinate clamp (kept out of line: rare)
template < int PC >
__device__ __noinline__ uint32_t
fetch4_clamped (const uint8_t * __restrict__ ref, int stride, int w, int h, int sx, int sy,
    int prec)
{
  uint32_t pk = 0;              // four u8 samples, returned in a register
  for (int e = 0; e < 4; e++)
    pk |= (uint32_t) fetch_ref < PC > (ref, stride, w, h, sx + e * (1 << prec), sy, prec) << (8 * e);
  return pk;
}

constexpr int kAccMargin = 3;   // a 4-pixel segment may stick out of the tile by 3 pixels

struct TileCtx {
  int x_lo, x_hi, y_lo, y_hi;
  int xfold_hi, yfold_hi;
};

// Generic item: border blocks (per-sample clamp) and picture-edge weight
// folding (accumulate_slow, schromotion8.c:673-693).  Rare, not tuned.
template < int PC >
__device__ __forceinline__ void
obmc_item_slow (const ObmcJob & job, const BlkInfo & bi, int row, int seg, const TileCtx & tc,
    const int *s_wx, const int *s_wy, int *acc)
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
      for (int e = 0; e < 4; e++)
        val[r][e] = fetch_ref < PC > (job.ref[r], job.ref_stride[r], job.w, job.h,
            bi.fx[r] + (4 * seg + e) * (1 << prec), bi.fy[r] + row * (1 << prec), prec);
    }
    for (int e = 0; e < 4; e++)
      pred[e] = mode == 3 ? (val[0][e] + val[1][e] + 1) >> 1 : (mode == 1 ? val[0][e] : val[1][e]);
  }
  int wy = s_wy[row];
  if (y < job.yoff)
    wy += s_wy[2 * job.yoff - row - 1];
  if (y >= tc.yfold_hi)
    wy += s_wy[2 * (job.yblen - job.yoff) - row - 1];
  int *arow = acc + (y - tc.y_lo) * kAccStride + kAccMargin - tc.x_lo;
  for (int e = 0; e < 4; e++) {
    const int x = xs + e, idx = 4 * seg + e;
    if (idx >= job.xblen)
      continue;
    int wx = s_wx[idx];
    if (x < job.xoff)
      wx += s_wx[2 * job.xoff - idx - 1];
    if (x >= tc.xfold_hi)
      wx += s_wx[2 * (job.xblen - job.xoff) - idx - 1];
    atomicAdd (arow + x, pred[e] * wx * wy);
  }
}

// Hot item: both sample windows inside the image, no weight folding.
// wx0[e] is 0 for padding pixels of a partial last segment.
template < int PC >
__device__ __forceinline__ void
obmc_item_fast (const ObmcJob & job, const BlkInfo & bi, int md, int row, int seg, int y, int xs,
    const TileCtx & tc, int wy0, const int *wx0, int *acc)
{
  constexpr int kStep = PC == 0 ? 1 : 2;
  const int mode = md & 3;
  int v0[4], v1[4];
  const int dcv = md >> 8;
  v0[0] = v0[1] = v0[2] = v0[3] = dcv;
  if (mode & 1)
    fetch4_inside < PC > (job.ref[0] + bi.off[0] + (row * kStep) * job.ref_stride[0]
        + seg * (4 * kStep), job.ref_stride[0], bi.wpk[0], v0);
#pragma unroll
  for (int e = 0; e < 4; e++)
    v1[e] = v0[e];              // one reference (or DC): avg (a, a) == a
  if (mode & 2) {
    fetch4_inside < PC > (job.ref[1] + bi.off[1] + (row * kStep) * job.ref_stride[1]
        + seg * (4 * kStep), job.ref_stride[1], bi.wpk[1], v1);
    if (!(mode & 1)) {
#pragma unroll
      for (int e = 0; e < 4; e++)
        v0[e] = v1[e];
    }
  }
  int *ap = acc + (y - tc.y_lo) * kAccStride + kAccMargin + (xs - tc.x_lo);
#pragma unroll
  for (int e = 0; e < 4; e++)
    atomicAdd (ap + e, ((v0[e] + v1[e] + 1) >> 1) * (wx0[e] * wy0));
}

// SLOW == false: the hot pass, handles the items that need neither clamping
// nor folding and returns true if this item was left for the slow pass.
// SLOW == true: the (rare) second pass over exactly those items.
template < int PC, bool SLOW >
__device__ __forceinline__ bool
obmc_item (const ObmcJob & job, const BlkInfo & bi, int row, int seg, const TileCtx & tc,
    int wy0, const int *wx0, const int *s_wx, const int *s_wy, int *acc)
{
  const int y = bi.by + row, xs = bi.bx + 4 * seg;
  if (y < tc.y_lo || y >= tc.y_hi || xs + 3 < tc.x_lo || xs >= tc.x_hi)
    return false;
  const int md = bi.mode_dc;
  const bool clamped = ((md >> 2) & md & 3) != 0;       // a reference in use needs clamping
  const bool fold = y < job.yoff || y >= tc.yfold_hi || xs < job.xoff || xs + 3 >= tc.xfold_hi;
  if constexpr (SLOW) {
    if (clamped || fold)
      obmc_item_slow < PC > (job, bi, row, seg, tc, s_wx, s_wy, acc);
    return false;
  } else {
    if (clamped || fold)
      return true;
    obmc_item_fast < PC > (job, bi, md, row, seg, y, xs, tc, wy0, wx0, acc);
    return false;
  }
}

// out = sat_u8 (residual + ((acc + 32) >> 6)) for one tile, 4 pixels per lane
__device__ __forceinline__ void
obmc_finish (const ObmcJob & job, const int *acc, int tid, int x_lo, int y_lo, int x_hi, int y_hi)
{
  // orc_rrshift6_add_s16_2d / _s32_2d on 4 pixels per lane
  for (int it = tid; it < kFTH * kFTW / 4; it += kThreads) {
    const int g = it % (kFTW / 4), yy = it / (kFTW / 4);
    const int x = x_lo + 4 * g, y = y_lo + yy;
    if (y >= y_hi || x >= x_hi)
      continue;
    const int *ap = acc + yy * kAccStride + kAccMargin + 4 * g;
    const int av[4] = { ap[0], ap[1], ap[2], ap[3] };
    const char *rrow = (const char *) job.residual + (size_t) y * job.residual_stride;
    uint8_t *orow = job.out + (size_t) y * job.out_stride + x;
    __attribute__ ((aligned (8))) int16_t res[4];
    const bool full = x + 4 <= job.w;
    if (job.res_bpp == 2) {
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

template < int PC >
__global__ __launch_bounds__ (kThreads)
void obmc_tile_kernel (const ObmcJob * __restrict__ jobs, int njobs)
{
  __shared__ int acc[kFTH * kAccStride];
  __shared__ int s_wx[kMaxBlk], s_wy[kMaxBlk];
  __shared__ BlkInfo s_blk[kBlkCap];
  __shared__ int s_cnt[4];

  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  const ObmcJob job = jobs[find_job (jobs, njobs, bid)];
  const int t = bid - job.tile_base;
  const int tx = t % job.tiles_x, ty = t / job.tiles_x;
  const int tid = threadIdx.x;
  const int x_lo = tx * kFTW, y_lo = ty * kFTH;
  const int x_hi = min (x_lo + kFTW, job.w), y_hi = min (y_lo + kFTH, job.h);

  for (int it = tid; it < kFTH * kAccStride; it += kThreads)
    acc[it] = 0;
  if (tid < job.xblen)
    s_wx[tid] = obmc_weight_1d (tid, job.xblen, job.xoff);
  if (tid >= 64 && tid - 64 < job.yblen)
    s_wy[tid - 64] = obmc_weight_1d (tid - 64, job.yblen, job.yoff);

  const int xblen = job.xblen, yblen = job.yblen, xbsep = job.xbsep, ybsep = job.ybsep;
  const int xoff = job.xoff, yoff = job.yoff, prec = job.prec;
  // blocks whose footprint meets the tile
  const int i_lo = max (0, floor_div (x_lo + xoff - xblen, xbsep) + 1);
  const int i_hi = min (job.nbx - 1, (x_hi - 1 + xoff) / xbsep);
  const int j_lo = max (0, floor_div (y_lo + yoff - yblen, ybsep) + 1);
  const int j_hi = min (job.nby - 1, (y_hi - 1 + yoff) / ybsep);
  const int nbi = i_hi - i_lo + 1, nbj = j_hi - j_lo + 1;
  const int nblk = nbi > 0 && nbj > 0 ? nbi * nbj : 0;
  const int nseg = (xblen + 3) >> 2;
  const int per_block = yblen * nseg;
  const int expx = 32 << prec;
  const int max_fast_x = (job.w - xblen) * (1 << prec), max_fast_y = (job.h - yblen) * (1 << prec);
  const int xfold_hi = job.nbx * xbsep - xoff, yfold_hi = job.nby * ybsep - yoff;
  // sample-grid geometry of one reference image
  const int gw = PC == 0 ? job.w - 1 : 2 * job.w - 2;   // last valid sample column
  const int gh = PC == 0 ? job.h - 1 : 2 * job.h - 2;
  constexpr int kStep = PC == 0 ? 1 : 2;        // samples per pixel step

  // lane -> (block slot, row, segment), fixed for the whole tile when a block
  // fits in the workgroup; C blocks are processed per pass
  const int C = per_block <= kThreads ? kThreads / per_block : 0;
  const int total_per_pass = C ? C * per_block : kThreads;
  int b_local = 0, row = 0, seg = 0;
  if (C) {
    b_local = tid / per_block;
    const int rem = tid - b_local * per_block;
    row = rem / nseg;
    seg = rem - row * nseg;
  }
  TileCtx tc;
  tc.x_lo = x_lo;
  tc.x_hi = x_hi;
  tc.y_lo = y_lo;
  tc.y_hi = y_hi;
  tc.xfold_hi = xfold_hi;
  tc.yfold_hi = yfold_hi;
  __syncthreads ();             // weights visible
  int wy0 = 0, wx0[4] = { 0, 0, 0, 0 };
  if (C && tid < total_per_pass) {
    wy0 = s_wy[row];
#pragma unroll
    for (int e = 0; e < 4; e++)
      wx0[e] = 4 * seg + e < xblen ? s_wx[4 * seg + e] : 0;
  }

  for (int chunk0 = 0; chunk0 < nblk; chunk0 += kBlkCap) {
    const int nb = min (kBlkCap, nblk - chunk0);
    if (tid < 4)
      s_cnt[tid] = 0;
    __syncthreads ();           // acc/weights ready; previous chunk's table consumed
    // ---- decode this chunk's motion vectors once per block (kBlkCap == kThreads:
    // one block per thread) and place them in the table sorted by prediction
    // mode, so that the lanes of a pass mostly take the same fetch branches ------
    static_assert (kBlkCap == kThreads, "one block per thread per chunk");
    BlkInfo info;
    int key = 0, rank = 0;
    const bool have = tid < nb;
    if (have) {
      const int b = tid;
      const int blk = chunk0 + b;
      const int bj = blk / nbi;
      const int i = i_lo + (blk - bj * nbi), jj = j_lo + bj;
      const uint8_t *mvp = job.mvs + (size_t) 20 * ((size_t) jj * job.nbx + i);
      const uint32_t flags = gload < uint32_t > (mvp);
      const uint32_t v01 = gload < uint32_t > (mvp + 12);
      const uint32_t v23 = gload < uint32_t > (mvp + 16);
      info.bx = xbsep * i - xoff;
      info.by = ybsep * jj - yoff;
      const int mode = flags & 3;
      const bool interior = i >= 1 && i < job.max_x_blocks && jj >= 1 && jj < job.max_y_blocks;
      int dc = job.comp == 0 ? (int16_t) (v01 & 0xffff)
          : job.comp == 1 ? (int16_t) (v01 >> 16) : (int16_t) (v23 & 0xffff);
      // get_dc_block stores a uint8_t; block_acc_dc multiplies a 16-bit parameter
      int p = interior ? (int) (int16_t) (dc + 128) : (int) (uint8_t) (dc + 128);
      int md = mode | (p << 8);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        int dx = r == 0 ? (int16_t) (v01 & 0xffff) : (int16_t) (v01 >> 16);
        int dy = r == 0 ? (int16_t) (v23 & 0xffff) : (int16_t) (v23 >> 16);
        dx >>= job.mv_shift_x;
        dy >>= job.mv_shift_y;
        const int fx = clampi (info.bx * (1 << prec) + dx, -expx, max_fast_x + expx - 1);
        const int fy = clampi (info.by * (1 << prec) + dy, -expx, max_fast_y + expx - 1);
        info.fx[r] = fx;
        info.fy[r] = fy;
        // first sample of the block on the reference's sample grid, and the
        // last one any of its pixels touches
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
        const bool inside = gx0 >= 0 && gy0 >= 0 && gx1 <= gw && gy1 <= gh;
        if (!inside)
          md |= 4 << r;
        info.off[r] = inside ? gy0 * job.ref_stride[r] + gx0 : 0;
        info.wpk[r] = wpk;
      }
      info.mode_dc = md;
      info.pad = 0;
      key = mode == 3 ? 0 : (mode == 1 ? 1 : (mode == 2 ? 2 : 3));
      rank = atomicAdd (&s_cnt[key], 1);
    }
    __syncthreads ();
    if (have) {
      int base = 0;
      for (int k = 0; k < key; k++)
        base += s_cnt[k];
      s_blk[base + rank] = info;
    }
    __syncthreads ();
    // ---- accumulate: one (block, row, 4-pixel segment) per lane per pass ------
    if (C) {
      if (tid < total_per_pass) {
        bool leftover = false;
        for (int b = b_local; b < nb; b += C)
          leftover |= obmc_item < PC, false > (job, s_blk[b], row, seg, tc, wy0, wx0, s_wx, s_wy, acc);
        if (leftover)
          for (int b = b_local; b < nb; b += C)
            obmc_item < PC, true > (job, s_blk[b], row, seg, tc, wy0, wx0, s_wx, s_wy, acc);
      }
    } else {
      // blocks larger than the workgroup: generic item decode, everything on the slow path
      for (int item = tid; item < nb * per_block; item += kThreads) {
        const int b = item / per_block;
        const int rem = item - b * per_block;
        const int r2 = rem / nseg, s2 = rem - r2 * nseg;
        const BlkInfo & bi = s_blk[b];
        const int y = bi.by + r2, xs = bi.bx + 4 * s2;
        if (y < y_lo || y >= y_hi || xs + 3 < x_lo || xs >= x_hi)
          continue;
        obmc_item_slow < PC > (job, bi, r2, s2, tc, s_wx, s_wy, acc);
      }
    }
  }
  __syncthreads ();

  obmc_finish (job, acc, tid, x_lo, y_lo, x_hi, y_hi);
}

template < int PC >
int
launch_one (hipStream_t stream, const ObmcJob * d_jobs, int njobs, int total_tiles, int variant)
{
  if (variant == 1)
    hipLaunchKernelGGL ((obmc_tile_kernel < PC >), dim3 (total_tiles), dim3 (kThreads), 0, stream,
        d_jobs, njobs);
  else
    hipLaunchKernelGGL ((obmc_kernel < PC, false >), dim3 (total_tiles), dim3 (kThreads), 0,
        stream, d_jobs, njobs);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "obmc launch: %s", hipGetErrorString (e));
  return 0;
}

}                               // namespace

// variant 0: per-pixel kernel (any weights), 64x4 tiles
// variant 1: LDS-accumulate tile kernel (default weights), 128x32 tiles
void
obmc_tiles (int variant, int w, int h, int xoff, int *tiles_x, int *tiles_y)
{
  (void) xoff;
  if (variant == 1) {
    *tiles_x = (w + kFTW - 1) / kFTW;
    *tiles_y = (h + kFTH - 1) / kFTH;
  } else {
    *tiles_x = (w + kTW - 1) / kTW;
    *tiles_y = (h + kTH - 1) / kTH;
  }
}

int
launch_obmc (hipStream_t stream, const ObmcJob * d_jobs, int njobs, int total_tiles, int prec,
    int variant)
{
  switch (prec == 0 ? 0 : (prec == 1 ? 1 : 2)) {
    case 0: return launch_one < 0 > (stream, d_jobs, njobs, total_tiles, variant);
    case 1: return launch_one < 1 > (stream, d_jobs, njobs, total_tiles, variant);
    default: return launch_one < 2 > (stream, d_jobs, njobs, total_tiles, variant);
  }
}

}                               // namespace schro
