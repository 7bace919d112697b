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

namespace schro {
namespace {

enum { K_ADD2_22, K_AVG11, K_MAS4, K_HAAR_HALF, K_HAAR_FULL, K_MAS8, K_MAS2 };

struct Step {
  int target;                   // 0: A (even / low half) updated from B, 1: B from A
  int kind;
  int off;                      // first neighbour index relative to i
  int sign;
  int c, rnd, sh;
};

// Synthesis step lists: schro_synth_ext_desl93 :1466, _53 :1542, _135 :1616,
// haar :1697-1764, _fidelity :1768, _daub97 :1894 (same table as
// oracle/oracle_wavelet.c, which is checked against the reference's kernels).
__host__ __device__ constexpr int
filter_nsteps (int f)
{
  return f == 6 ? 4 : 2;
}

__host__ __device__ constexpr Step
filter_step (int f, int k)
{
  switch (f) {
    case 0:
      return k == 0 ? Step {0, K_ADD2_22, -1, -1, 0, 2, 2}
                    : Step {1, K_MAS4, -1, +1, 0, 8, 4};
    case 1:
      return k == 0 ? Step {0, K_ADD2_22, -1, -1, 0, 2, 2}
                    : Step {1, K_AVG11, 0, +1, 0, 1, 1};
    case 2:
      return k == 0 ? Step {0, K_MAS4, -2, -1, 0, 16, 5}
                    : Step {1, K_MAS4, -1, +1, 0, 8, 4};
    case 3:
    case 4:
      return k == 0 ? Step {0, K_HAAR_HALF, 0, -1, 0, 1, 1}
                    : Step {1, K_HAAR_FULL, 0, +1, 0, 0, 0};
    case 5:
      return k == 0 ? Step {1, K_MAS8, -3, +1, 0, 128, 8}
                    : Step {0, K_MAS8, -4, +1, 1, 127, 8};
    default:
      return k == 0 ? Step {0, K_MAS2, -1, -1, 1817, 2048, 12}
           : k == 1 ? Step {1, K_MAS2, 0, -1, 3616, 2048, 12}
           : k == 2 ? Step {0, K_MAS2, -1, +1, 217, 2048, 12}
                    : Step {1, K_MAS2, 0, +1, 6497, 2048, 12};
  }
}

// lifting halo in sub-band samples (both directions)
__host__ __device__ constexpr int
filter_halo (int f)
{
  return f == 0 ? 2 : f == 1 ? 1 : f == 2 ? 3 : f == 5 ? 7 : f == 6 ? 2 : 0;
}

// 0 none, 1 wrapping (x+1)>>1 (orc_interleave2_rrshift1_*), 2 avgs(x,0)
// (orc_haar_synth_rrshift1_int_*)
__host__ __device__ constexpr int
filter_shift (int f)
{
  return (f == 3 || f == 5) ? 0 : (f == 4 ? 2 : 1);
}

__host__ __device__ constexpr int
kind_ntaps (int kind)
{
  return kind == K_MAS4 ? 4 : kind == K_MAS8 ? 8
       : (kind == K_HAAR_HALF || kind == K_HAAR_FULL) ? 1 : 2;
}

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

// ---- horizontal lifting step on sample quads (i .. i+3), i a multiple of 4 -------
// CLAMP == false: the region lies inside the picture; every quad of the row is
// computed from 64/128-bit LDS reads.  Taps that fall off the region at its two
// ends read neighbouring LDS words (always inside the array: rows 2H.. are never
// the first or last row) and only spoil halo samples no useful output depends on.
// LAST == true: the filter's final step; the updated quad and its partner quad
// from the other half are interleaved, rounded and stored straight to memory
// (no LDS write-back, no separate output pass).
template < typename T, int F, int K, int RP, int RC, int H, int HC, int UR, bool CLAMP, bool LAST >
__device__ __forceinline__ void
horizontal_step (T (*lds)[2 * RC], int tid, int hlo, int hhi, int rows_here, const IwtJob & job,
    int y0, int c0, int nc)
{
  struct __attribute__ ((aligned (4 * sizeof (T)))) T4 { T v[4]; };
  constexpr Step st = filter_step (F, K);
  constexpr int NT = kind_ntaps (st.kind);
  constexpr int SH = filter_shift (F);
  constexpr int QLO = LAST ? HC / 4 : 0;
  constexpr int QHI = LAST ? (RC - HC) / 4 - 1 : RC / 4 - 1;
  constexpr int NQ = QHI - QLO + 1;
  // neighbour window: samples i+off .. i+off+NT+2, fetched as aligned 4-sample words
  constexpr int FIRST = floor4 (st.off);
  constexpr int NW = (st.off + NT + 2 - FIRST) / 4 + 1;
  static_assert (RC % 4 == 0 && HC % 4 == 0, "quads need 4-aligned halves");
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
      const int c = c0 + i;
      if (c >= nc || c + 3 < 0)
        continue;
      // the partner quad o[i..i+3] lies inside the neighbour window for every filter
      static_assert (-st.off >= 0 && -st.off <= NT - 1, "partner quad outside the tap window");
      T out[8];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const T ov = s[k - st.off];
        out[2 * k] = out_round < T, SH > (st.target ? ov : dv.v[k]);
        out[2 * k + 1] = out_round < T, SH > (st.target ? dv.v[k] : ov);
      }
      T *dst = (T *) ((char *) job.dst + (size_t) (y0 + yy) * job.dst_stride) + 2 * c;
      if ((job.flags & 2) && c >= 0 && c + 4 <= nc) {
        if constexpr (sizeof (T) == 2) {
          uint4 pk;
          pk.x = (uint16_t) out[0] | ((uint32_t) (uint16_t) out[1] << 16);
          pk.y = (uint16_t) out[2] | ((uint32_t) (uint16_t) out[3] << 16);
          pk.z = (uint16_t) out[4] | ((uint32_t) (uint16_t) out[5] << 16);
          pk.w = (uint16_t) out[6] | ((uint32_t) (uint16_t) out[7] << 16);
          *reinterpret_cast < uint4 * >(dst) = pk;
        } else {
          reinterpret_cast < int4 * >(dst)[0] = make_int4 (out[0], out[1], out[2], out[3]);
          reinterpret_cast < int4 * >(dst)[1] = make_int4 (out[4], out[5], out[6], out[7]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (c + k >= 0 && c + k < nc) {
            dst[2 * k] = out[2 * k];
            dst[2 * k + 1] = out[2 * k + 1];
          }
      }
    }
  }
}

template < typename T, int F, int K, bool CLAMP, typename LDS >
__device__ __forceinline__ void
hstep (LDS lds, int tid, int hlo, int hhi, int rows_here, const IwtJob & job, int y0, int c0, int nc)
{
  typedef Geo < T, F > G;
  constexpr bool LAST = K == filter_nsteps (F) - 1;
  horizontal_step < T, F, K, G::RP, G::RC, G::H, G::HC, G::UR, CLAMP, LAST > (lds, tid, hlo, hhi,
      rows_here, job, y0, c0, nc);
  if constexpr (!LAST)
    __syncthreads ();
}

template < typename T, int F >
__global__ __launch_bounds__ (kThreads)
void iiwt_level_kernel (const IwtJob * __restrict__ jobs, int njobs)
{
  typedef Geo < T, F > G;
  constexpr int RP = G::RP, RC = G::RC, H = G::H, HC = G::HC, UR = G::UR, UC = G::UC;
  constexpr int VL = 8 / sizeof (T);    // samples per 8-byte vector
  constexpr int NG = RC / VL;           // 8-byte groups per half row
  __shared__ __attribute__ ((aligned (16))) T lds[2 * RP][2 * RC];

  const int tid = threadIdx.x;
  int j = 0;
  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  while (j + 1 < njobs && bid >= jobs[j + 1].tile_base)
    j++;
  const IwtJob job = jobs[j];
  const int t = bid - job.tile_base;
  const int tx = t % job.tiles_x, ty = t / job.tiles_x;
  const int nr = job.h >> 1, nc = job.w >> 1;
  const int r0 = ty * UR - H;   // sub-band row of region row pair 0
  const int c0 = tx * UC - HC;  // sub-band column of region column 0
  // region-local index range that exists in the picture
  const int vlo = max (0, -r0), vhi = min (RP - 1, nr - 1 - r0);
  const int hlo = max (0, -c0), hhi = min (RC - 1, nc - 1 - c0);

  // ---- stage the four sub-bands of the region in LDS ----------------------
  if (job.flags & 1) {
    // sub-band index and LDS half are compile-time per load (no runtime-indexed
    // job fields), all loads are issued before the first LDS write
    constexpr int NPS = (RP * NG + kThreads - 1) / kThreads;    // loads per thread per sub-band
    uint2 v[4][NPS];
#pragma unroll
    for (int sb = 0; sb < 4; sb++) {
      const char *base = (const char *) job.sb[sb];
      const int stride = job.sb_stride[sb];
#pragma unroll
      for (int n = 0; n < NPS; n++) {
        int it = min (tid + n * kThreads, RP * NG - 1);
        int g = it % NG;
        int rp = it / NG;
        int r = clampi (r0 + rp, 0, nr - 1);
        int c = clampi (c0 + g * VL, 0, nc - VL);
        v[sb][n] = *reinterpret_cast < const uint2 * >(base + (size_t) r * stride
            + (size_t) c * sizeof (T));
      }
    }
#pragma unroll
    for (int sb = 0; sb < 4; sb++) {
#pragma unroll
      for (int n = 0; n < NPS; n++) {
        int it = tid + n * kThreads;
        if (it < RP * NG) {
          int g = it % NG;
          int rp = it / NG;
          *reinterpret_cast < uint2 * >(&lds[2 * rp + (sb >> 1)][(sb & 1) * RC + g * VL]) = v[sb][n];
        }
      }
    }
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
              ((const T *) (base + (size_t) r * stride))[cc];
      }
    }
  }
  __syncthreads ();

  // ---- vertical lifting steps (A = even rows, B = odd rows) ----------------
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

  // ---- horizontal lifting steps on the rows this tile outputs; the last one
  // interleaves, rounds and stores ---------------------------------------------
  const int y0 = 2 * (r0 + H);  // first output row of the tile
  const int rows_here = min (2 * UR, job.h - y0);
  if (hlo > 0 || hhi < RC - 1) {
    hstep < T, F, 0, true > (lds, tid, hlo, hhi, rows_here, job, y0, c0, nc);
    hstep < T, F, 1, true > (lds, tid, hlo, hhi, rows_here, job, y0, c0, nc);
    if constexpr (filter_nsteps (F) == 4) {
      hstep < T, F, 2, true > (lds, tid, hlo, hhi, rows_here, job, y0, c0, nc);
      hstep < T, F, 3, true > (lds, tid, hlo, hhi, rows_here, job, y0, c0, nc);
    }
  } else {
    hstep < T, F, 0, false > (lds, tid, hlo, hhi, rows_here, job, y0, c0, nc);
    hstep < T, F, 1, false > (lds, tid, hlo, hhi, rows_here, job, y0, c0, nc);
    if constexpr (filter_nsteps (F) == 4) {
      hstep < T, F, 2, false > (lds, tid, hlo, hhi, rows_here, job, y0, c0, nc);
      hstep < T, F, 3, false > (lds, tid, hlo, hhi, rows_here, job, y0, c0, nc);
    }
  }
}

template < typename T, int F >
int
launch_one (hipStream_t stream, const IwtJob * d_jobs, int njobs, int total_tiles)
{
  hipLaunchKernelGGL ((iiwt_level_kernel < T, F >), dim3 (total_tiles), dim3 (kThreads), 0,
      stream, d_jobs, njobs);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "iiwt launch: %s", hipGetErrorString (e));
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

}                               // namespace schro
