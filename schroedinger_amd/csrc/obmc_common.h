// obmc_common.h -- device helpers shared by the OBMC kernels (obmc.hip, obmc_stage.hip).
#pragma once

#include "schro_hip_internal.h"

namespace schro {
namespace {

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
  return 1 + (6 * x + offset - 1) / (2 * offset - 1);   // once per thread; no table round trip
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
// PC 0: plain plane.  PC 1: half-pel image (tiled).  PC 2: 1/4- or 1/8-pel bilinear
// of four half-pel samples (orc_combine4_nxm_u8, schroorc.orc:1635-1662; the
// avg2 / copy special cases of schroframe.c:2306-2350 are the same formula).
// ps / cb: the sample's width shift and the component's byte in it (pair images, schro_hip_internal.h)
template < int PC >
__device__ __forceinline__ int
fetch_ref (const uint8_t * __restrict__ ref, int stride, int w, int h, int sx, int sy, int prec, int ps, int cb)
{
  if constexpr (PC == 0) {
    int X = clampi (sx, 0, w - 1), Y = clampi (sy, 0, h - 1);
    return gload < uint8_t > (ref + (size_t) Y * stride + X);
  } else if constexpr (PC == 1) {
    int X = clampi (sx, 0, 2 * w - 2), Y = clampi (sy, 0, 2 * h - 2);
    return gload < uint8_t > (ref + hp_offset (X, Y, stride, ps, cb));
  } else {
    int x8 = prec == 2 ? sx * 2 : sx, y8 = prec == 2 ? sy * 2 : sy;
    int hx = x8 >> 2, hy = y8 >> 2, rx = x8 & 3, ry = y8 & 3;
    int X0 = clampi (hx, 0, 2 * w - 2), X1 = clampi (hx + 1, 0, 2 * w - 2);
    int Y0 = clampi (hy, 0, 2 * h - 2), Y1 = clampi (hy + 1, 0, 2 * h - 2);
    int p00 = gload < uint8_t > (ref + hp_offset (X0, Y0, stride, ps, cb)), p01 = gload < uint8_t > (ref + hp_offset (X1, Y0, stride, ps, cb));
    int p10 = gload < uint8_t > (ref + hp_offset (X0, Y1, stride, ps, cb)), p11 = gload < uint8_t > (ref + hp_offset (X1, Y1, stride, ps, cb));
    int v = (4 - ry) * ((4 - rx) * p00 + rx * p01) + ry * ((4 - rx) * p10 + rx * p11);
    return (v + 8) >> 4;
  }
}

// get_block's clamped fetch origin of reference r, in 1/2^prec pel (schromotion8.c:303-335)
__device__ __forceinline__ void
mv_origin (const ObmcJob & job, int bx, int by, uint32_t v01, uint32_t v23, int r, int *fx, int *fy)
{
  const int prec = job.prec, expx = 32 << prec;
  const int max_fast_x = (job.w - job.xblen) * (1 << prec), max_fast_y = (job.h - job.yblen) * (1 << prec);
  int dx = r == 0 ? (int16_t) (v01 & 0xffff) : (int16_t) (v01 >> 16);
  int dy = r == 0 ? (int16_t) (v23 & 0xffff) : (int16_t) (v23 >> 16);
  dx >>= job.mv_shift_x;
  dy >>= job.mv_shift_y;
  *fx = clampi (bx * (1 << prec) + dx, -expx, max_fast_x + expx - 1);
  *fy = clampi (by * (1 << prec) + dy, -expx, max_fast_y + expx - 1);
}

}                               // namespace
}                               // namespace schro
