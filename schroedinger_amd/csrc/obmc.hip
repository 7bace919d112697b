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
    return ref[(size_t) Y * stride + X];
  } else if constexpr (PC == 1) {
    int X = clampi (sx, 0, 2 * w - 2), Y = clampi (sy, 0, 2 * h - 2);
    return ref[(size_t) Y * stride + X];
  } else {
    int x8 = prec == 2 ? sx * 2 : sx, y8 = prec == 2 ? sy * 2 : sy;
    int hx = x8 >> 2, hy = y8 >> 2, rx = x8 & 3, ry = y8 & 3;
    int X0 = clampi (hx, 0, 2 * w - 2), X1 = clampi (hx + 1, 0, 2 * w - 2);
    int Y0 = clampi (hy, 0, 2 * h - 2), Y1 = clampi (hy + 1, 0, 2 * h - 2);
    const uint8_t *r0 = ref + (size_t) Y0 * stride, *r1 = ref + (size_t) Y1 * stride;
    int p00 = r0[X0], p01 = r0[X1], p10 = r1[X0], p11 = r1[X1];
    int v = (4 - ry) * ((4 - rx) * p00 + rx * p01) + ry * ((4 - rx) * p10 + rx * p11);
    return (v + 8) >> 4;
  }
}

template < int PC, bool SIMPLE >
__global__ __launch_bounds__ (kThreads)
void obmc_kernel (const ObmcJob * __restrict__ jobs, int njobs)
{
  __shared__ int s_wx[kMaxBlk], s_wy[kMaxBlk];

  int j = 0;
  while (j + 1 < njobs && (int) blockIdx.x >= jobs[j + 1].tile_base)
    j++;
  const ObmcJob job = jobs[j];
  const int t = blockIdx.x - job.tile_base;
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
      const uint32_t flags = *reinterpret_cast < const uint32_t * >(mvp);
      const uint32_t v01 = *reinterpret_cast < const uint32_t * >(mvp + 12);
      const uint32_t v23 = *reinterpret_cast < const uint32_t * >(mvp + 16);
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
    res = ((const int16_t *) ((const char *) job.residual + (size_t) py * job.residual_stride))[px];
  else
    res = (int16_t) ((const int32_t *) ((const char *) job.residual +
            (size_t) py * job.residual_stride))[px];
  t1 = (int16_t) (res + t1);
  job.out[(size_t) py * job.out_stride + px] = (uint8_t) clampi (t1, 0, 255);
}

template < int PC, bool SIMPLE >
int
launch_one (hipStream_t stream, const ObmcJob * d_jobs, int njobs, int total_tiles)
{
  hipLaunchKernelGGL ((obmc_kernel < PC, SIMPLE >), dim3 (total_tiles), dim3 (kThreads), 0,
      stream, d_jobs, njobs);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "obmc launch: %s", hipGetErrorString (e));
  return 0;
}

}                               // namespace

void
obmc_tile_geometry (int *tw, int *th)
{
  *tw = kTW;
  *th = kTH;
}

int
launch_obmc (hipStream_t stream, const ObmcJob * d_jobs, int njobs, int total_tiles, int prec,
    int simple_weight)
{
  int pc = prec == 0 ? 0 : (prec == 1 ? 1 : 2);
  if (simple_weight) {
    switch (pc) {
      case 0: return launch_one < 0, true > (stream, d_jobs, njobs, total_tiles);
      case 1: return launch_one < 1, true > (stream, d_jobs, njobs, total_tiles);
      default: return launch_one < 2, true > (stream, d_jobs, njobs, total_tiles);
    }
  }
  switch (pc) {
    case 0: return launch_one < 0, false > (stream, d_jobs, njobs, total_tiles);
    case 1: return launch_one < 1, false > (stream, d_jobs, njobs, total_tiles);
    default: return launch_one < 2, false > (stream, d_jobs, njobs, total_tiles);
  }
}

}                               // namespace schro
