// iiwt.hip -- one level of the 2-D inverse integer lifting wavelet, all seven
// Dirac filters, s16 and s32, for gfx950.
//
// What it computes: schro_wavelet_inverse_transform_2d
// (schroedinger/schrowaveletorc.c:121-188) for one level view, i.e. the
// filters schro_iiwt_desl_9_3 :1475, _5_3 :1551, _13_5 :1625, _haar0/1 :1697,
// _fidelity :1844, _daub_9_7 :1996 and their _s32 twins :2055-2667, with the
// kernel arithmetic of schroedinger/schroorc.orc (16-bit wrap points for s16,
// 32-bit wrap for s32).
//
// How (MI355X-first, not the reference's row-skewed in-place schedule):
//   * the level is cut into tiles; one 256-thread workgroup owns one tile and
//     keeps the tile PLUS its lifting halo in LDS as [row][low half | high half];
//   * the four sub-bands of the tile are fetched with 8-byte coalesced loads
//     (all loads of a thread are issued before the first LDS write);
//   * every lifting step is an in-place LDS pass: vertical steps work on
//     column pairs (one 32/64-bit LDS access = two samples), horizontal steps
//     on row pairs; neighbour indices clamp to the picture inside the same
//     array, exactly like extend_N_M / CLAMP(row, ...) in the reference;
//   * the finished rows are interleaved, rounded ((x+1)>>1 where the filter
//     has an output shift) and stored with 16-byte coalesced stores.
//   The level reads LL from the previous level's compact output and HL/LH/HH
//   from the coefficient frame, and writes a compact image: one read and one
//   write of every sample per level, nothing in place, so tiles never race.
//
// Bound: HBM.  Algorithmic bytes per output sample per level: 2 * sizeof(T).

#include "schro_hip_internal.h"
#include "iiwt_steps.h"

namespace schro {
namespace {

// fidelity taps: stage 1 (c == 0) and stage 2 (c == 1), schrowaveletorc.c:1771-1773
__device__ constexpr int
fid_tap (int which, int k)
{
  constexpr int s1[8] = { -2, 10, -25, 81, 81, -25, 10, -2 };
  constexpr int s2[8] = { 8, -21, 46, -161, -161, 46, -21, 8 };
  return which ? s2[k] : s1[k];
}

template < typename T > struct Ar;
template <> struct Ar < int16_t > {
  typedef int16_t T;
  struct __attribute__ ((aligned (4))) T2 { T x, y; };
  static __device__ __forceinline__ T wrap (int v) { return (T) v; }
  static __device__ __forceinline__ int mul (T a, int c) { return (int) a * c; }
  static __device__ __forceinline__ int add32 (int a, int b) { return a + b; }
  static __device__ __forceinline__ int sub32 (int a, int b) { return a - b; }
  static __device__ __forceinline__ T avg (T a, T b) { return (T) (((int) a + (int) b + 1) >> 1); }
};
template <> struct Ar < int32_t > {
  typedef int32_t T;
  struct __attribute__ ((aligned (8))) T2 { T x, y; };
  static __device__ __forceinline__ T wrap (int v) { return v; }
  static __device__ __forceinline__ int mul (T a, int c) { return (int) ((unsigned) a * (unsigned) c); }
  static __device__ __forceinline__ int add32 (int a, int b) { return (int) ((unsigned) a + (unsigned) b); }
  static __device__ __forceinline__ int sub32 (int a, int b) { return (int) ((unsigned) a - (unsigned) b); }
  static __device__ __forceinline__ T avg (T a, T b) { return (T) (((long long) a + (long long) b + 1) >> 1); }
};

// value of one lifting term from its neighbours (same arithmetic as
// oracle_wavelet_tmpl.h lift_term, i.e. schroorc.orc's opcode lists)
template < typename T, int F, int K >
__device__ __forceinline__ T
lift_term (const T * s)
{
  typedef Ar < T > A;
  constexpr Step st = filter_step (F, K);
  if constexpr (st.kind == K_ADD2_22) {
    T t = A::wrap (A::add32 (s[0], s[1]));
    t = A::wrap (A::add32 (t, 2));
    return (T) (t >> 2);
  } else if constexpr (st.kind == K_AVG11) {
    return A::avg (s[0], s[1]);
  } else if constexpr (st.kind == K_MAS4) {
    T t1 = A::wrap (A::add32 (s[1], s[2]));
    int t3 = A::mul (t1, 9);
    T t2 = A::wrap (A::add32 (s[0], s[3]));
    t3 = A::sub32 (t3, t2);
    t3 = A::add32 (t3, st.rnd);
    t3 >>= st.sh;
    return A::wrap (t3);
  } else if constexpr (st.kind == K_HAAR_HALF) {
    return A::avg (s[0], 0);
  } else if constexpr (st.kind == K_HAAR_FULL) {
    return s[0];
  } else if constexpr (st.kind == K_MAS8) {
    int x = st.rnd;
#pragma unroll
    for (int k = 0; k < 8; k++)
      x = A::add32 (x, A::mul (s[k], fid_tap (st.c, k)));
    return A::wrap (x >> 8);
  } else {
    T t1 = A::wrap (A::add32 (s[0], s[1]));
    int t2 = A::mul (t1, st.c);
    t2 = A::add32 (t2, st.rnd);
    t2 >>= st.sh;
    return A::wrap (t2);
  }
}

template < typename T, int F, int K >
__device__ __forceinline__ T
lift_apply (T d, const T * s)
{
  typedef Ar < T > A;
  constexpr Step st = filter_step (F, K);
  T t = lift_term < T, F, K > (s);
  if constexpr (st.sign > 0)
    return A::wrap (A::add32 (d, t));
  else
    return A::wrap (A::sub32 (d, t));
}

template < typename T, int F > struct Geo {
  static constexpr int RP = 32;                         // region row pairs
  static constexpr int H = filter_halo (F);
  static constexpr int HC = (H + 3) & ~3;               // keeps 8-byte alignment of loads
  // useful columns per half per tile: a tile row of output is 2*UC samples and must be
  // whole 128-byte lines, otherwise neighbouring tiles (on other XCDs) each write a
  // partial line
  static constexpr int UC = sizeof (T) == 2 ? 128 : 64;
  static constexpr int RC = UC + 2 * HC;                // region columns per half
  static constexpr int UR = RP - 2 * H;                 // useful row pairs per tile
};

__device__ __forceinline__ int
clampi (int x, int lo, int hi)
{
  return min (max (x, lo), hi);
}

constexpr int kThreads = 256;

constexpr int
cmax (int a, int b)
{
  return a > b ? a : b;
}

// ---- vertical lifting step: 4 columns (one 64/128-bit LDS access) per item ----
// CLAMP == false: the region lies inside the picture, neighbour rows are taken as
// they are and only rows whose taps stay inside the region are computed (the
// skipped rows are halo whose values no useful output depends on).
template < typename T, int F, int K, int RP, int RC, bool CLAMP >
__device__ __forceinline__ void
vertical_step (T (*lds)[2 * RC], int tid, int vlo, int vhi)
{
  struct __attribute__ ((aligned (4 * sizeof (T)))) T4 { T v[4]; };
  constexpr Step st = filter_step (F, K);
  constexpr int NT = kind_ntaps (st.kind);
  constexpr int IPR = 2 * RC / 4;       // items per row
  constexpr int LO = cmax (0, -st.off), HI = RP - 1 - cmax (0, st.off + NT - 1);
#pragma unroll 2
  for (int it = tid; it < RP * IPR; it += kThreads) {
    const int cp = it % IPR;
    const int rp = it / IPR;
    if (CLAMP ? (rp < vlo || rp > vhi) : (rp < LO || rp > HI))
      continue;
    T4 tap[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int rr = CLAMP ? clampi (rp + st.off + t, vlo, vhi) : rp + st.off + t;
      tap[t] = reinterpret_cast < const T4 * >(&lds[2 * rr + 1 - st.target][0])[cp];
    }
    T4 *dp = reinterpret_cast < T4 * >(&lds[2 * rp + st.target][0]) + cp;
    T4 d = *dp;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      T s[NT];
#pragma unroll
      for (int t = 0; t < NT; t++)
        s[t] = tap[t].v[e];
      d.v[e] = lift_apply < T, F, K > (d.v[e], s);
    }
    *dp = d;
  }
}

template < typename T, int SH >
__device__ __forceinline__ T
out_round (T a)
{
  typedef Ar < T > A;
  if constexpr (SH == 1)
    return (T) (A::wrap (A::add32 (a, 1)) >> 1);        // orc_interleave2_rrshift1_*: add wraps
  else if constexpr (SH == 2)
    return A::avg (a, 0);                               // orc_haar_synth_rrshift1_int_*: avgs
  else
    return a;
}

constexpr int
floor4 (int a)
{
  return a >= 0 ? a / 4 * 4 : -((-a + 3) / 4 * 4);
}

constexpr int
ceil4 (int a)
{
  return -floor4 (-a);
}

__device__ __forceinline__ int
floor_div2 (int a)
{
  return a >> 1;                // arithmetic shift: floor for negatives too
}

// ---- where the last lifting step puts its interleaved, rounded output ------------
// to memory (the level's destination image)
template < typename T > struct GlobalSink {
  char *dst;
  int dst_stride;               // bytes
  int y0;                       // output row of tile row 0
  int c0;                       // sub-band column of region column 0
  int nc;                       // sub-band columns of the level
  bool vec;                     // destination rows are 16-byte aligned

  __device__ __forceinline__ void store (int yy, int i, const T * out) const
  {
    const int c = c0 + i;
    if (c >= nc || c + 3 < 0)
      return;
    T *p = (T *) (dst + (size_t) (y0 + yy) * dst_stride) + 2 * c;
    if (vec && c >= 0 && c + 4 <= nc) {
      if constexpr (sizeof (T) == 2) {
        u32x4 pk;
        pk.x = (uint16_t) out[0] | ((uint32_t) (uint16_t) out[1] << 16);
        pk.y = (uint16_t) out[2] | ((uint32_t) (uint16_t) out[3] << 16);
        pk.z = (uint16_t) out[4] | ((uint32_t) (uint16_t) out[5] << 16);
        pk.w = (uint16_t) out[6] | ((uint32_t) (uint16_t) out[7] << 16);
        gstore < u32x4 > (p, pk);
      } else {
        gstore < u32x4 > (p, (u32x4) { (uint32_t) out[0], (uint32_t) out[1], (uint32_t) out[2], (uint32_t) out[3] });
        gstore < u32x4 > (p + 4, (u32x4) { (uint32_t) out[4], (uint32_t) out[5], (uint32_t) out[6], (uint32_t) out[7] });
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (c + k >= 0 && c + k < nc) {
          gstore < T > (p + 2 * k, out[2 * k]);
          gstore < T > (p + 2 * k + 1, out[2 * k + 1]);
        }
    }
  }
};

// into the LL quadrant of the next finer level's LDS region (fused levels): output
// sample (y, x) of this level is LL (row y, column x) of the finer level, which
// lives at finer_lds[2 * (y - y_org)][x - x_org]
template < typename T > struct LdsSink {
  T *base;                      // &finer_lds[0][0]
  int row_stride;               // elements between two LL rows (2 LDS rows)
  int y_org, x_org;             // finer region origin (sub-band row pair / column)
  int y_lo, y_hi, x_lo, x_hi;   // LL samples the finer region holds and the picture has
  int y0, c0;                   // this level: output row of tile row 0, region column origin

  __device__ __forceinline__ void store (int yy, int i, const T * out) const
  {
    const int y = y0 + yy;
    if (y < y_lo || y >= y_hi)
      return;
    const int x0 = 2 * (c0 + i);
    T *p = base + (y - y_org) * row_stride + (x0 - x_org);
    if (x0 >= x_lo && x0 + 8 <= x_hi) {
      struct __attribute__ ((aligned (4 * sizeof (T)))) T4 { T v[4]; };
      T4 a = { {out[0], out[1], out[2], out[3]} }, b = { {out[4], out[5], out[6], out[7]} };
      reinterpret_cast < T4 * >(p)[0] = a;
      reinterpret_cast < T4 * >(p)[1] = b;
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (x0 + k >= x_lo && x0 + k < x_hi)
          p[k] = out[k];
    }
  }
};

// ---- horizontal lifting step on sample quads (i .. i+3), i a multiple of 4 -------
// CLAMP == false: the region lies inside the picture; every quad of the row is
// computed from 64/128-bit LDS reads.  Taps that fall off the region at its two
// ends read neighbouring LDS words and only spoil halo samples no useful output
// depends on.
// LAST == true: the filter's final step; the updated quad and its partner quad
// from the other half are interleaved, rounded and handed to the sink (no LDS
// write-back, no separate output pass); only quads QLO..QHI are visited.
template < typename T, int F, int K, int RP, int RC, int H, int QLO_LAST, int QHI_LAST, bool CLAMP,
    bool LAST, typename SINK >
__device__ __forceinline__ void
horizontal_step (T (*lds)[2 * RC], int tid, int hlo, int hhi, int rows_here, const SINK & sink)
{
  struct __attribute__ ((aligned (4 * sizeof (T)))) T4 { T v[4]; };
  constexpr Step st = filter_step (F, K);
  constexpr int NT = kind_ntaps (st.kind);
  constexpr int SH = filter_shift (F);
  constexpr int UR = RP - 2 * H;
  constexpr int QLO = LAST ? QLO_LAST : 0;
  constexpr int QHI = LAST ? QHI_LAST : RC / 4 - 1;
  constexpr int NQ = QHI - QLO + 1;
  // neighbour window: samples i+off .. i+off+NT+2, fetched as aligned 4-sample words
  constexpr int FIRST = floor4 (st.off);
  constexpr int NW = (st.off + NT + 2 - FIRST) / 4 + 1;
  static_assert (RC % 4 == 0, "quads need 4-aligned halves");
#pragma unroll 2
  for (int it = tid; it < 2 * UR * NQ; it += kThreads) {
    const int q = QLO + it % NQ;
    const int yy = it / NQ;
    const int i = 4 * q;
    if (yy >= rows_here)
      continue;
    if (CLAMP && (i > hhi || i + 3 < hlo))
      continue;
    T *row = &lds[2 * H + yy][0];
    T *d = row + (st.target ? RC : 0);
    const T *o = row + (st.target ? 0 : RC);
    T s[NT + 3];
    if constexpr (CLAMP) {
#pragma unroll
      for (int t = 0; t < NT + 3; t++)
        s[t] = o[clampi (i + st.off + t, hlo, hhi)];
    } else {
      T4 w[NW];
#pragma unroll
      for (int m = 0; m < NW; m++)
        w[m] = reinterpret_cast < const T4 * >(o + i + FIRST)[m];
#pragma unroll
      for (int t = 0; t < NT + 3; t++) {
        const int e = t + st.off - FIRST;
        s[t] = w[e >> 2].v[e & 3];
      }
    }
    T4 dv = *reinterpret_cast < const T4 * >(d + i);
#pragma unroll
    for (int k = 0; k < 4; k++)
      dv.v[k] = lift_apply < T, F, K > (dv.v[k], s + k);
    if constexpr (!LAST) {
      *reinterpret_cast < T4 * >(d + i) = dv;
    } else {
      // the partner quad o[i..i+3] lies inside the neighbour window for every filter
      static_assert (-st.off >= 0 && -st.off <= NT - 1, "partner quad outside the tap window");
      T out[8];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const T ov = s[k - st.off];
        out[2 * k] = out_round < T, SH > (st.target ? ov : dv.v[k]);
        out[2 * k + 1] = out_round < T, SH > (st.target ? dv.v[k] : ov);
      }
      sink.store (yy, i, out);
    }
  }
}

template < typename T, int F, int K, int RP, int RC, int H, int QLO, int QHI, bool CLAMP, typename SINK >
__device__ __forceinline__ void
hstep (T (*lds)[2 * RC], int tid, int hlo, int hhi, int rows_here, const SINK & sink)
{
  constexpr bool LAST = K == filter_nsteps (F) - 1;
  horizontal_step < T, F, K, RP, RC, H, QLO, QHI, CLAMP, LAST, SINK > (lds, tid, hlo, hhi,
      rows_here, sink);
  if constexpr (!LAST)
    __syncthreads ();
}

// All lifting passes of one level on a staged region: vertical steps in place,
// horizontal steps on the rows whose vertical result is valid, the last one
// feeding the sink.  (r0, c0): sub-band row pair / column of the region origin.
template < typename T, int F, int RP, int RC, int QLO, int QHI, typename SINK >
__device__ __forceinline__ void
lift_region (T (*lds)[2 * RC], int tid, int r0, int c0, int nr, int nc, const SINK & sink)
{
  constexpr int H = filter_halo (F);
  constexpr int UR = RP - 2 * H;
  // region-local index range that exists in the picture
  const int vlo = max (0, -r0), vhi = min (RP - 1, nr - 1 - r0);
  const int hlo = max (0, -c0), hhi = min (RC - 1, nc - 1 - c0);

  if (vlo > 0 || vhi < RP - 1) {        // region sticks out of the picture: clamp rows
    vertical_step < T, F, 0, RP, RC, true > (lds, tid, vlo, vhi);
    __syncthreads ();
    vertical_step < T, F, 1, RP, RC, true > (lds, tid, vlo, vhi);
    __syncthreads ();
    if constexpr (filter_nsteps (F) == 4) {
      vertical_step < T, F, 2, RP, RC, true > (lds, tid, vlo, vhi);
      __syncthreads ();
      vertical_step < T, F, 3, RP, RC, true > (lds, tid, vlo, vhi);
      __syncthreads ();
    }
  } else {
    vertical_step < T, F, 0, RP, RC, false > (lds, tid, vlo, vhi);
    __syncthreads ();
    vertical_step < T, F, 1, RP, RC, false > (lds, tid, vlo, vhi);
    __syncthreads ();
    if constexpr (filter_nsteps (F) == 4) {
      vertical_step < T, F, 2, RP, RC, false > (lds, tid, vlo, vhi);
      __syncthreads ();
      vertical_step < T, F, 3, RP, RC, false > (lds, tid, vlo, vhi);
      __syncthreads ();
    }
  }

  const int rows_here = min (2 * UR, 2 * nr - 2 * (r0 + H));    // rows inside the picture
  if (hlo > 0 || hhi < RC - 1) {
    hstep < T, F, 0, RP, RC, H, QLO, QHI, true > (lds, tid, hlo, hhi, rows_here, sink);
    hstep < T, F, 1, RP, RC, H, QLO, QHI, true > (lds, tid, hlo, hhi, rows_here, sink);
    if constexpr (filter_nsteps (F) == 4) {
      hstep < T, F, 2, RP, RC, H, QLO, QHI, true > (lds, tid, hlo, hhi, rows_here, sink);
      hstep < T, F, 3, RP, RC, H, QLO, QHI, true > (lds, tid, hlo, hhi, rows_here, sink);
    }
  } else {
    hstep < T, F, 0, RP, RC, H, QLO, QHI, false > (lds, tid, hlo, hhi, rows_here, sink);
    hstep < T, F, 1, RP, RC, H, QLO, QHI, false > (lds, tid, hlo, hhi, rows_here, sink);
    if constexpr (filter_nsteps (F) == 4) {
      hstep < T, F, 2, RP, RC, H, QLO, QHI, false > (lds, tid, hlo, hhi, rows_here, sink);
      hstep < T, F, 3, RP, RC, H, QLO, QHI, false > (lds, tid, hlo, hhi, rows_here, sink);
    }
  }
}

// Issue the 8-byte loads of one sub-band of a region (coordinates clamped into the
// picture; cells outside it are never read back) and, later, put them in LDS.
template < typename T, int RP, int RC, int NPS >
__device__ __forceinline__ void
band_load (uint2 * v, const void *base_, int stride, int tid, int r0, int c0, int nr, int nc)
{
  constexpr int VL = 8 / sizeof (T), NG = RC / VL;
  const char *base = (const char *) base_;
#pragma unroll
  for (int n = 0; n < NPS; n++) {
    int it = min (tid + n * kThreads, RP * NG - 1);
    int g = it % NG;
    int rp = it / NG;
    int rr = clampi (r0 + rp, 0, nr - 1);
    int c = clampi (c0 + g * VL, 0, nc - VL);
    const u32x2 q = gload < u32x2 > (base + (size_t) rr * stride + (size_t) c * sizeof (T));
    v[n] = make_uint2 (q.x, q.y);
  }
}

template < typename T, int RP, int RC, int NPS >
__device__ __forceinline__ void
band_store (T (*lds)[2 * RC], const uint2 * v, int sb, int tid)
{
  constexpr int VL = 8 / sizeof (T), NG = RC / VL;
#pragma unroll
  for (int n = 0; n < NPS; n++) {
    int it = tid + n * kThreads;
    if (it < RP * NG) {
      int g = it % NG;
      int rp = it / NG;
      *reinterpret_cast < uint2 * >(&lds[2 * rp + (sb >> 1)][(sb & 1) * RC + g * VL]) = v[n];
    }
  }
}

template < typename T, int RP, int RC >
constexpr int
band_nps ()
{
  return (RP * (RC / (8 / (int) sizeof (T))) + kThreads - 1) / kThreads;
}

template < typename T, int F >
__global__ __launch_bounds__ (kThreads)
void iiwt_level_kernel (const IwtJob * __restrict__ jobs, int njobs)
{
  typedef Geo < T, F > G;
  constexpr int RP = G::RP, RC = G::RC, H = G::H, HC = G::HC, UR = G::UR, UC = G::UC;
  constexpr int NPS = band_nps < T, RP, RC > ();
  __shared__ __attribute__ ((aligned (16))) T lds[2 * RP][2 * RC];

  const int tid = threadIdx.x;
  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  const IwtJob job = jobs[find_job (jobs, njobs, bid)];
  const int t = bid - job.tile_base;
  const int tx = t % job.tiles_x, ty = t / job.tiles_x;
  const int nr = job.h >> 1, nc = job.w >> 1;
  const int r0 = ty * UR - H;   // sub-band row of region row pair 0
  const int c0 = tx * UC - HC;  // sub-band column of region column 0

  // ---- stage the four sub-bands of the region in LDS ----------------------
  if (job.flags & 1) {
    // all loads are issued before the first LDS write
    uint2 v[4][NPS];
    band_load < T, RP, RC, NPS > (v[0], job.sb[0], job.sb_stride[0], tid, r0, c0, nr, nc);
    band_load < T, RP, RC, NPS > (v[1], job.sb[1], job.sb_stride[1], tid, r0, c0, nr, nc);
    band_load < T, RP, RC, NPS > (v[2], job.sb[2], job.sb_stride[2], tid, r0, c0, nr, nc);
    band_load < T, RP, RC, NPS > (v[3], job.sb[3], job.sb_stride[3], tid, r0, c0, nr, nc);
    band_store < T, RP, RC, NPS > (lds, v[0], 0, tid);
    band_store < T, RP, RC, NPS > (lds, v[1], 1, tid);
    band_store < T, RP, RC, NPS > (lds, v[2], 2, tid);
    band_store < T, RP, RC, NPS > (lds, v[3], 3, tid);
  } else {
#pragma unroll
    for (int sb = 0; sb < 4; sb++) {
      const char *base = (const char *) job.sb[sb];
      const int stride = job.sb_stride[sb];
      for (int it = tid; it < RP * RC; it += kThreads) {
        int c = it % RC;
        int rp = it / RC;
        int r = r0 + rp, cc = c0 + c;
        if (r >= 0 && r < nr && cc >= 0 && cc < nc)
          lds[2 * rp + (sb >> 1)][(sb & 1) * RC + c] =
              gload < T > ((const T *) (base + (size_t) r * stride) + cc);
      }
    }
  }
  __syncthreads ();

  GlobalSink < T > sink;
  sink.dst = (char *) job.dst;
  sink.dst_stride = job.dst_stride;
  sink.y0 = 2 * (r0 + H);
  sink.c0 = c0;
  sink.nc = nc;
  sink.vec = (job.flags & 2) != 0;
  lift_region < T, F, RP, RC, HC / 4, (RC - HC) / 4 - 1 > (lds, tid, r0, c0, nr, nc, sink);
}

// ---- fused levels ---------------------------------------------------------------
// One workgroup produces a level-0 tile from the sub-bands of NL levels: the coarser
// levels are computed for exactly the LL region (plus lifting halo) the next finer
// level's region needs and land directly in that region's LL quadrant in LDS.  The LL
// images of the fused levels never exist in memory, and NL launches become one.
// Region geometry per fused level L (0 = finest), all compile-time:
//   rows:    RP_L = RP_{L-1}/2 + 1 + 2H row pairs, origin floor (r0_{L-1} / 2) - H
//   columns: origin tx * (UC0 >> L) + O_L with O_L a multiple of 4 (8-byte loads stay
//            aligned) far enough left for the halo, RC_L columns up to the halo on the right
template < typename T, int F, int L > struct FGeo {
  typedef FGeo < T, F, L - 1 > P;
  static constexpr int H = filter_halo (F);
  static constexpr int RP = P::RP / 2 + 1 + 2 * H;
  static constexpr int CB = P::CB / 2;                  // column pitch of tiles at this level
  static constexpr int S = P::O / 2;                    // first LL column the finer region needs
  static constexpr int O = floor4 (S - H);
  static constexpr int RC = ceil4 (S + P::RC / 2 + H - O);
};
template < typename T, int F > struct FGeo < T, F, 0 > {
  typedef Geo < T, F > G;
  static constexpr int H = G::H;
  static constexpr int RP = G::RP;
  static constexpr int CB = G::UC;
  static constexpr int O = -G::HC;
  static constexpr int RC = G::RC;
};

struct IwtFusedJob {
  const void *band[3][3];       // [level][HL, LH, HH]: element (0,0) of each sub-band
  int bstride[3];               // bytes between sub-band rows, per level
  int ll_stride;
  const void *ll;               // LL of the coarsest fused level
  void *dst;
  int dst_stride;
  int w, h;                     // level-0 output size
  int tiles_x;
  int tile_base;
  int flags;                    // bit1: dst 16-byte aligned
};

template < typename T, int F, int NL >
__global__ __launch_bounds__ (kThreads)
void iiwt_fused_kernel (const IwtFusedJob * __restrict__ jobs, int njobs)
{
  typedef FGeo < T, F, 0 > G0;
  typedef FGeo < T, F, 1 > G1;
  typedef FGeo < T, F, NL == 3 ? 2 : 1 > G2;    // only used when NL == 3
  constexpr int H = G0::H, UR0 = G0::RP - 2 * H;
  constexpr int NPS0 = band_nps < T, G0::RP, G0::RC > ();
  constexpr int NPS1 = band_nps < T, G1::RP, G1::RC > ();
  constexpr int NPS2 = band_nps < T, G2::RP, G2::RC > ();
  __shared__ __attribute__ ((aligned (16))) T lds0[2 * G0::RP][2 * G0::RC];
  __shared__ __attribute__ ((aligned (16))) T lds1[2 * G1::RP][2 * G1::RC];
  __shared__ __attribute__ ((aligned (16))) T lds2[NL == 3 ? 2 * G2::RP : 1][2 * G2::RC];

  const int tid = threadIdx.x;
  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  const IwtFusedJob job = jobs[find_job (jobs, njobs, bid)];
  const int t = bid - job.tile_base;
  const int tx = t % job.tiles_x, ty = t / job.tiles_x;
  // per level: sub-band rows / columns of the picture, region origin
  const int nr0 = job.h >> 1, nc0 = job.w >> 1;
  const int nr1 = job.h >> 2, nc1 = job.w >> 2;
  const int nr2 = job.h >> 3, nc2 = job.w >> 3;
  const int r00 = ty * UR0 - H, c00 = tx * G0::CB + G0::O;
  const int r01 = floor_div2 (r00) - H, c01 = tx * G1::CB + G1::O;
  const int r02 = floor_div2 (r01) - H, c02 = tx * G2::CB + G2::O;

  // ---- every sub-band load of all fused levels goes out first ------------------
  uint2 v0[3][NPS0], v1[3][NPS1], v2[4][NPS2];
#pragma unroll
  for (int b = 0; b < 3; b++)
    band_load < T, G0::RP, G0::RC, NPS0 > (v0[b], job.band[0][b], job.bstride[0], tid, r00, c00, nr0, nc0);
#pragma unroll
  for (int b = 0; b < 3; b++)
    band_load < T, G1::RP, G1::RC, NPS1 > (v1[b], job.band[1][b], job.bstride[1], tid, r01, c01, nr1, nc1);
  if constexpr (NL == 3) {
#pragma unroll
    for (int b = 0; b < 3; b++)
      band_load < T, G2::RP, G2::RC, NPS2 > (v2[b + 1], job.band[2][b], job.bstride[2], tid, r02, c02, nr2, nc2);
    band_load < T, G2::RP, G2::RC, NPS2 > (v2[0], job.ll, job.ll_stride, tid, r02, c02, nr2, nc2);
  } else {
    band_load < T, G1::RP, G1::RC, NPS1 > (v2[0], job.ll, job.ll_stride, tid, r01, c01, nr1, nc1);
  }
#pragma unroll
  for (int b = 0; b < 3; b++)
    band_store < T, G0::RP, G0::RC, NPS0 > (lds0, v0[b], b + 1, tid);
#pragma unroll
  for (int b = 0; b < 3; b++)
    band_store < T, G1::RP, G1::RC, NPS1 > (lds1, v1[b], b + 1, tid);
  if constexpr (NL == 3) {
#pragma unroll
    for (int b = 0; b < 4; b++)
      band_store < T, G2::RP, G2::RC, NPS2 > (lds2, v2[b], b, tid);
  } else {
    static_assert (NPS2 == NPS1 || NL == 3, "LL staging of the 2-level form reuses v2[0]");
    band_store < T, G1::RP, G1::RC, NPS1 > (lds1, v2[0], 0, tid);
  }
  __syncthreads ();

  if constexpr (NL == 3) {
    // level 2 -> LL quadrant of the level-1 region
    LdsSink < T > s2;
    s2.base = &lds1[0][0];
    s2.row_stride = 2 * 2 * G1::RC;
    s2.y_org = r01;
    s2.x_org = c01;
    s2.y_lo = max (r01, 0);
    s2.y_hi = min (r01 + G1::RP, nr1);
    s2.x_lo = max (c01, 0);
    s2.x_hi = min (c01 + G1::RC, nc1);
    s2.y0 = 2 * (r02 + H);
    s2.c0 = c02;
    lift_region < T, F, G2::RP, G2::RC, 0, G2::RC / 4 - 1 > (lds2, tid, r02, c02, nr2, nc2, s2);
    __syncthreads ();
  }
  {
    // level 1 -> LL quadrant of the level-0 region
    LdsSink < T > s1;
    s1.base = &lds0[0][0];
    s1.row_stride = 2 * 2 * G0::RC;
    s1.y_org = r00;
    s1.x_org = c00;
    s1.y_lo = max (r00, 0);
    s1.y_hi = min (r00 + G0::RP, nr0);
    s1.x_lo = max (c00, 0);
    s1.x_hi = min (c00 + G0::RC, nc0);
    s1.y0 = 2 * (r01 + H);
    s1.c0 = c01;
    lift_region < T, F, G1::RP, G1::RC, 0, G1::RC / 4 - 1 > (lds1, tid, r01, c01, nr1, nc1, s1);
    __syncthreads ();
  }
  GlobalSink < T > sink;
  sink.dst = (char *) job.dst;
  sink.dst_stride = job.dst_stride;
  sink.y0 = 2 * (r00 + H);
  sink.c0 = c00;
  sink.nc = nc0;
  sink.vec = (job.flags & 2) != 0;
  constexpr int HC0 = -G0::O;
  lift_region < T, F, G0::RP, G0::RC, HC0 / 4, (G0::RC - HC0) / 4 - 1 > (lds0, tid, r00, c00, nr0, nc0, sink);
}

template < typename T, int F >
int
launch_one (hipStream_t stream, const IwtJob * d_jobs, int njobs, int total_tiles)
{
  SCHRO_LAUNCH ((iiwt_level_kernel < T, F >), dim3 (total_tiles), dim3 (kThreads), 0,
      stream, d_jobs, njobs);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "iiwt launch: %s", hipGetErrorString (e));
  return 0;
}

template < typename T, int F, int NL >
constexpr size_t
fused_lds_bytes ()
{
  size_t b = sizeof (T) * 4 * FGeo < T, F, 0 >::RP * FGeo < T, F, 0 >::RC
      + sizeof (T) * 4 * FGeo < T, F, 1 >::RP * FGeo < T, F, 1 >::RC;
  if (NL == 3)
    b += sizeof (T) * 4 * FGeo < T, F, 2 >::RP * FGeo < T, F, 2 >::RC;
  return b;
}

// how many of the finest levels one launch can fuse for this filter / sample type:
// bounded by the 64 KB of LDS a workgroup may declare; the 8-tap fidelity filter's
// halo of 7 makes the coarse regions mostly halo, so it is never fused
template < typename T, int F >
constexpr int
fused_max_levels ()
{
  if (F == 5)
    return 0;
  if (fused_lds_bytes < T, F, 3 > () <= 65536)
    return 3;
  if (fused_lds_bytes < T, F, 2 > () <= 65536)
    return 2;
  return 0;
}

template < typename T, int F >
int
launch_fused_one (hipStream_t stream, const void *d_jobs, int njobs, int total_tiles, int nl)
{
  constexpr int MAXL = fused_max_levels < T, F > ();
  bool launched = false;
  if constexpr (MAXL >= 3) {
    if (nl == 3) {
      SCHRO_LAUNCH ((iiwt_fused_kernel < T, F, 3 >), dim3 (total_tiles), dim3 (kThreads), 0,
          stream, (const IwtFusedJob *) d_jobs, njobs);
      launched = true;
    }
  }
  if constexpr (MAXL >= 2) {
    if (nl == 2) {
      SCHRO_LAUNCH ((iiwt_fused_kernel < T, F, 2 >), dim3 (total_tiles), dim3 (kThreads), 0,
          stream, (const IwtFusedJob *) d_jobs, njobs);
      launched = true;
    }
  }
  if (!launched)
    return set_error (SCHRO_HIP_EINVAL, "fused iiwt: %d levels not available for filter %d", nl, F);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "fused iiwt launch: %s", hipGetErrorString (e));
  return 0;
}

template < typename T >
int
fused_levels_of (int filter)
{
  switch (filter) {
    case 0: return fused_max_levels < T, 0 > ();
    case 1: return fused_max_levels < T, 1 > ();
    case 2: return fused_max_levels < T, 2 > ();
    case 3: return fused_max_levels < T, 3 > ();
    case 4: return fused_max_levels < T, 4 > ();
    case 6: return fused_max_levels < T, 6 > ();
  }
  return 0;
}

template < typename T >
int
launch_filter (hipStream_t stream, const IwtJob * d_jobs, int njobs, int total_tiles, int filter)
{
  switch (filter) {
    case 0: return launch_one < T, 0 > (stream, d_jobs, njobs, total_tiles);
    case 1: return launch_one < T, 1 > (stream, d_jobs, njobs, total_tiles);
    case 2: return launch_one < T, 2 > (stream, d_jobs, njobs, total_tiles);
    case 3: return launch_one < T, 3 > (stream, d_jobs, njobs, total_tiles);
    case 4: return launch_one < T, 4 > (stream, d_jobs, njobs, total_tiles);
    case 5: return launch_one < T, 5 > (stream, d_jobs, njobs, total_tiles);
    case 6: return launch_one < T, 6 > (stream, d_jobs, njobs, total_tiles);
  }
  return set_error (SCHRO_HIP_EINVAL, "wavelet filter index %d out of range", filter);
}

template < typename T >
int
launch_fused_filter (hipStream_t stream, const void *d_jobs, int njobs, int total_tiles, int filter, int nl)
{
  switch (filter) {
    case 0: return launch_fused_one < T, 0 > (stream, d_jobs, njobs, total_tiles, nl);
    case 1: return launch_fused_one < T, 1 > (stream, d_jobs, njobs, total_tiles, nl);
    case 2: return launch_fused_one < T, 2 > (stream, d_jobs, njobs, total_tiles, nl);
    case 3: return launch_fused_one < T, 3 > (stream, d_jobs, njobs, total_tiles, nl);
    case 4: return launch_fused_one < T, 4 > (stream, d_jobs, njobs, total_tiles, nl);
    case 5: return launch_fused_one < T, 5 > (stream, d_jobs, njobs, total_tiles, nl);
    case 6: return launch_fused_one < T, 6 > (stream, d_jobs, njobs, total_tiles, nl);
  }
  return set_error (SCHRO_HIP_EINVAL, "wavelet filter index %d out of range", filter);
}

template < typename T >
void
geometry (int filter, int *uc, int *ur)
{
  switch (filter) {
    case 0: *uc = Geo < T, 0 >::UC; *ur = Geo < T, 0 >::UR; break;
    case 1: *uc = Geo < T, 1 >::UC; *ur = Geo < T, 1 >::UR; break;
    case 2: *uc = Geo < T, 2 >::UC; *ur = Geo < T, 2 >::UR; break;
    case 3: *uc = Geo < T, 3 >::UC; *ur = Geo < T, 3 >::UR; break;
    case 4: *uc = Geo < T, 4 >::UC; *ur = Geo < T, 4 >::UR; break;
    case 5: *uc = Geo < T, 5 >::UC; *ur = Geo < T, 5 >::UR; break;
    default: *uc = Geo < T, 6 >::UC; *ur = Geo < T, 6 >::UR; break;
  }
}

}                               // namespace

void
iiwt_tile_geometry (int filter, int bpp, int *useful_cols, int *useful_row_pairs)
{
  if (bpp == 2)
    geometry < int16_t > (filter, useful_cols, useful_row_pairs);
  else
    geometry < int32_t > (filter, useful_cols, useful_row_pairs);
}

int
launch_iiwt_level (hipStream_t stream, const IwtJob * d_jobs, int njobs, int total_tiles,
    int filter, int bpp)
{
  if (bpp == 2)
    return launch_filter < int16_t > (stream, d_jobs, njobs, total_tiles, filter);
  return launch_filter < int32_t > (stream, d_jobs, njobs, total_tiles, filter);
}

size_t
iiwt_fused_job_size (void)
{
  return sizeof (IwtFusedJob);
}

// (the LDS-fused group is a measured-slower form: built into the experiments library only, schro_hip_internal.h)
int
iiwt_fused_max_levels (int filter, int bpp)
{
#ifdef SCHRO_HIP_EXPERIMENTS
  return bpp == 2 ? fused_levels_of < int16_t > (filter) : fused_levels_of < int32_t > (filter);
#else
  (void) filter;
  (void) bpp;
  return 0;
#endif
}

// fills one fused job (host side); levels 0 .. nl-1 of `src` are fused, `ll` is the LL
// image of level nl-1 (the coefficient frame itself when nl == depth)
void
iiwt_fused_job_fill (void *job_, const void *src, int src_stride, int bpp, int nl, const void *ll,
    int ll_stride, void *dst, int dst_stride, int w, int h, int tiles_x, int tile_base)
{
  IwtFusedJob *j = (IwtFusedJob *) job_;
  const char *base = (const char *) src;
  for (int l = 0; l < 3; l++) {
    const int lv = l < nl ? l : nl - 1;
    const int vstride = src_stride << lv;
    const int wl = w >> lv;
    // level view {w >> l, h >> l, stride << l}, sub-band positions schroparams.c:319-352
    j->band[l][0] = base + (size_t) (wl / 2) * bpp;                     // HL
    j->band[l][1] = base + vstride;                                     // LH
    j->band[l][2] = base + vstride + (size_t) (wl / 2) * bpp;           // HH
    j->bstride[l] = vstride * 2;
  }
  j->ll = ll;
  j->ll_stride = ll_stride;
  j->dst = dst;
  j->dst_stride = dst_stride;
  j->w = w;
  j->h = h;
  j->tiles_x = tiles_x;
  j->tile_base = tile_base;
  j->flags = ((((uintptr_t) dst | (uintptr_t) dst_stride) & 15) == 0) ? 2 : 0;
}

int
launch_iiwt_fused (hipStream_t stream, const void *d_jobs, int njobs, int total_tiles, int filter,
    int bpp, int nl)
{
#ifdef SCHRO_HIP_EXPERIMENTS
  if (bpp == 2)
    return launch_fused_filter < int16_t > (stream, d_jobs, njobs, total_tiles, filter, nl);
  return launch_fused_filter < int32_t > (stream, d_jobs, njobs, total_tiles, filter, nl);
#else
  (void) stream, (void) d_jobs, (void) njobs, (void) total_tiles, (void) filter, (void) bpp, (void) nl;
  return set_error (SCHRO_HIP_EUNSUPPORTED, "the LDS-fused wavelet group is built into the experiments library only");
#endif
}

}                               // namespace schro
