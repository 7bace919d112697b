// iiwt_reg.hip -- one level of the 2-D inverse lifting wavelet on s16 coefficients with
// the whole tile in registers: no LDS, no workgroup barrier.
//
// What it computes: the same as iiwt.hip (schro_wavelet_inverse_transform_2d,
// schroedinger/schrowaveletorc.c:121-188, s16 kernels of schroorc.orc) for the filters
// whose lifting halo is small: DD(9,7), LeGall(5,3), DD(13,7), Haar0/1, Daub(9,7).
//
// Why a second form: stamped timelines of the LDS kernel (profiles/r01_pmc_notes.md)
// show a workgroup spending half of its life in LDS lifting passes that four waves per
// SIMD execute at VALU issue rate, with its loads long finished and the next
// workgroup's not yet issued.  Here
//   * one WAVE owns a tile of 248 sub-band columns x (12 - 2H) row pairs plus halo;
//     lane l holds, for each of the 12 region row pairs, four adjacent columns of all
//     four sub-bands: 8 packed dwords per row pair, 96 VGPRs, loaded with 48
//     independent 8-byte global loads that are all in flight together;
//   * vertical lifting is lane-local on packed 16-bit pairs (v_pk_add_u16,
//     v_pk_ashrrev_i16, v_pk_mad_u16: exactly the Orc programs' 16-bit wrap points);
//     the 9*(b+c) - (a+d) steps, whose products need more than 16 bits, are split as
//     t = 2^sh * (t >> sh) + (t & (2^sh - 1)) so that each half stays inside 16 bits
//     and the result is the exact 32-bit value;
//   * horizontal lifting takes its neighbours from the adjacent lanes with DPP
//     wave shifts (one v_mov_b32_dpp per neighbour dword) and v_alignbit_b32 for the
//     odd sample offsets; lane 0 and lane 63 are the column halo;
//   * picture edges: the reference clamps neighbour indices inside the same array at
//     every step.  Rows: the top tile starts H row pairs above the picture and a tile at
//     the bottom has 1 .. H row pairs below it; each case is its own instantiation, so
//     the clamp is a compile-time register choice.  Columns: the first / last lane
//     inside the picture substitutes its own edge sample for the neighbour dwords (two
//     v_cndmask per fetched dword, only in the tiles that touch an edge);
//   * the finished rows are rounded, interleaved (v_perm_b32) and stored as 16 bytes
//     per lane, 992 contiguous bytes per wave and row.
// Waves are independent, so while one computes the others' loads and stores keep the
// memory pipe busy: that overlap is what the barrier-phased kernel lacked.
//
// Bound: HBM.  Algorithmic bytes per output sample: 2 B read + 2 B written; the row
// halo (2H of 12 row pairs) is re-read through L2.
//
// r04 -- the combine form (MODE bit 2, level 0 only).  The residual picture was written by this kernel and read
// once, by the OBMC finish (or the intra convert): 2 x 199 MB of a step of 8 x 2160p whose other traffic is
// 1 GB, and priced it costs more than its bytes -- OBMC without its residual reads takes 0.174 instead of
// 0.221 ms per step (the reads sit on every tile's critical path and push the reference planes out of the
// caches), the wavelet without its stores 0.041 + 0.026 instead of 0.067 + 0.037.  So the stages swap: the OBMC
// launch writes its prediction ((acc + 32) >> 6, a u8 plane) and THIS kernel's last step adds it:
// out = sat_u8 (residual + prediction) -- schro_motion_render's orc_rrshift6_add_s16_2d, schromotion8.c:852-857
// -- or + 128 for a picture without references (orc_offsetconvert_u8_s16, schrovirtframe.c:1689-1720).  The
// residual never exists in memory; per output sample this launch reads 2 + 1 B and writes 1 B.

#include "schro_hip_internal.h"
#include "iiwt_steps.h"

namespace schro {
namespace {

#define IWT_LOAD8(p) gload < u32x2 > (p)


// r04, the chain form (below): the intermediate LL images are written by one workgroup and read by others
// of the SAME launch, possibly on another XCD (each XCD has its own L2).  The STORES are agent-scope atomics
// (relaxed: global_store ... sc1, written through to the memory side); the loads are ordinary: a consumer reads
// a line of an LL image only after the whole producer tile rows that hold it have counted themselves done --
// the images' rows are whole 128-byte lines (plane_iiwt.cpp) -- so no cache of its XCD can hold an older copy from
// this launch, and every launch starts with clean caches.
// (Loads as agent-scope atomics too -- no argument needed -- cost the transform of 8 x 2160p 0.03 ms.)

__device__ __forceinline__ void
coh_store16 (char *p, u32x4 o)
{
  __hip_atomic_store ((SCHRO_GLOBAL uint64_t *) p, (uint64_t) o.x | ((uint64_t) o.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store ((SCHRO_GLOBAL uint64_t *) (p + 8), (uint64_t) o.z | ((uint64_t) o.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef short s16x2 __attribute__ ((ext_vector_type (2)));
typedef uint32_t P;             // two packed s16 samples

constexpr int kRegRP = 12;      // region row pairs held by a wave (bandwidth-bound levels)
constexpr int kRegUC = 4 * 62;  // useful columns per half per wave (lanes 1 .. 62)
constexpr int kRegThreads = 256;

__device__ __forceinline__ s16x2 S (P x) { return __builtin_bit_cast (s16x2, x); }
__device__ __forceinline__ P U (s16x2 x) { return __builtin_bit_cast (P, x); }

// value from the lane below / above (wave-wide shift; lane 0 / 63 get 0)
__device__ __forceinline__ P lane_prev (P x) { return __builtin_amdgcn_update_dpp (0u, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ P lane_next (P x) { return __builtin_amdgcn_update_dpp (0u, x, 0x130, 0xf, 0xf, false); }

__device__ __forceinline__ int
clampi (int x, int lo, int hi)
{
  return min (max (x, lo), hi);
}

constexpr int
cmin (int a, int b)
{
  return a < b ? a : b;
}

constexpr int
cmax (int a, int b)
{
  return a > b ? a : b;
}

// d +/- term (taps), two samples at once, with the s16 wrap points of schroorc.orc
template < int F, int K >
__device__ __forceinline__ P
lift_apply_pk (P d, const P * s)
{
  constexpr Step st = filter_step (F, K);
  s16x2 term;
  if constexpr (st.kind == K_ADD2_22) {
    term = ((S (s[0]) + S (s[1])) + (short) 2) >> 2;
  } else if constexpr (st.kind == K_AVG11) {
    // (a + b + 1) >> 1 without overflow
    term = (S (s[0]) | S (s[1])) - ((S (s[0]) ^ S (s[1])) >> 1);
  } else if constexpr (st.kind == K_MAS4) {
    // (9 * t1 - t2 + rnd) >> sh with t = 2^sh * a + b: 9 a1 - a2 + ((9 b1 - b2 + rnd) >> sh)
    constexpr uint32_t M = ((1u << st.sh) - 1) * 0x00010001u;
    const s16x2 t1 = S (s[1]) + S (s[2]), t2 = S (s[0]) + S (s[3]);
    const s16x2 a1 = t1 >> st.sh, a2 = t2 >> st.sh;
    const s16x2 b1 = S (U (t1) & M), b2 = S (U (t2) & M);
    const s16x2 q = (b1 * (short) 9 + (short) st.rnd - b2) >> st.sh;
    term = a1 * (short) 9 + q - a2;
  } else if constexpr (st.kind == K_HAAR_HALF) {
    term = S (s[0]) - (S (s[0]) >> 1);  // (a + 1) >> 1 without overflow
  } else if constexpr (st.kind == K_HAAR_FULL) {
    term = S (s[0]);
  } else {
    static_assert (st.kind == K_MAS2, "the 8-tap fidelity filter is not built in this form");
    // the product needs 32 bits: unpack, multiply, repack
    const P t = U (S (s[0]) + S (s[1]));
    const int lo = (int) (int16_t) (t & 0xffff), hi = (int) t >> 16;
    const int rl = (lo * st.c + st.rnd) >> st.sh, rh = (hi * st.c + st.rnd) >> st.sh;
    term = S (((uint32_t) rl & 0xffffu) | ((uint32_t) rh << 16));
  }
  if constexpr (st.sign > 0)
    return U (S (d) + term);
  else
    return U (S (d) - term);
}

struct Rng { int lo, hi; };

// Rows of step k's target array that have to be computed so that rows [H, RP - H) of
// both arrays are final after the last step; LO..HI are the rows that exist.
template < int F, int RP, int LO, int HI >
constexpr Rng
vert_rows (int k)
{
  constexpr int H = filter_halo (F);
  Rng need[2] = { {H, RP - H - 1}, {H, RP - H - 1} };
  for (int s = filter_nsteps (F) - 1; s >= 0; s--) {
    const Step st = filter_step (F, s);
    const int X = st.target, Y = 1 - X;
    const Rng rx = need[X];
    if (s == k)
      return rx;
    const int lo = cmax (LO, rx.lo + st.off), hi = cmin (HI, rx.hi + st.off + kind_ntaps (st.kind) - 1);
    need[Y] = Rng { cmin (need[Y].lo, lo), cmax (need[Y].hi, hi) };
  }
  return Rng { 0, -1 };
}

// rows of array `which` (0 even rows, 1 odd rows) that are read at all
template < int F, int RP, int LO, int HI >
constexpr Rng
load_rows (int which)
{
  constexpr int H = filter_halo (F);
  Rng need[2] = { {H, RP - H - 1}, {H, RP - H - 1} };
  for (int s = filter_nsteps (F) - 1; s >= 0; s--) {
    const Step st = filter_step (F, s);
    const int X = st.target, Y = 1 - X;
    const Rng rx = need[X];
    const int lo = cmax (LO, rx.lo + st.off), hi = cmin (HI, rx.hi + st.off + kind_ntaps (st.kind) - 1);
    need[Y] = Rng { cmin (need[Y].lo, lo), cmax (need[Y].hi, hi) };
  }
  return need[which];
}

// one vertical lifting step on the register tile: E = even rows (LL | HL), O = odd rows
template < int F, int K, int RP, int LO, int HI >
__device__ __forceinline__ void
vstep (P (&E)[RP][4], P (&O)[RP][4])
{
  constexpr Step st = filter_step (F, K);
  constexpr int NT = kind_ntaps (st.kind);
  constexpr Rng rg = vert_rows < F, RP, LO, HI > (K);
#pragma unroll
  for (int i = 0; i < RP; i++) {
    if (i < rg.lo || i > rg.hi)
      continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      P s[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const int idx = cmin (cmax (i + st.off + t, LO), HI);
        s[t] = st.target ? E[idx][j] : O[idx][j];
      }
      if (st.target)
        O[i][j] = lift_apply_pk < F, K > (O[i][j], s);
      else
        E[i][j] = lift_apply_pk < F, K > (E[i][j], s);
    }
  }
}

// the pair of samples starting at sample p of the window W = dwords of samples -4 .. 7
template < int p >
__device__ __forceinline__ P
pair_at (const P * W)
{
  constexpr int idx = p + 4;
  static_assert (idx >= 0 && idx + 1 <= 11, "tap outside the neighbour window");
  if constexpr (idx % 2 == 0)
    return W[idx / 2];
  else
    return __builtin_amdgcn_alignbit (W[(idx + 1) / 2], W[(idx - 1) / 2], 16);
}

template < int F, int K, int Q, int T, int NT >
__device__ __forceinline__ void
gather_taps (const P * W, P * s)
{
  if constexpr (T < NT) {
    constexpr Step st = filter_step (F, K);
    s[T] = pair_at < 2 * Q + st.off + T > (W);
    gather_taps < F, K, Q, T + 1, NT > (W, s);
  }
}

// one horizontal lifting step on one row: row[0..1] = low half samples 0..3 of this lane,
// row[2..3] = high half; neighbours come from the adjacent lanes
template < int F, int K, bool HEDGE >
__device__ __forceinline__ void
hstep (P (&row)[4], bool is_first, bool is_last)
{
  constexpr Step st = filter_step (F, K);
  constexpr int NT = kind_ntaps (st.kind);
  const P y0 = st.target ? row[0] : row[2], y1 = st.target ? row[1] : row[3];
  P p0 = lane_prev (y0), p1 = lane_prev (y1), n0 = lane_next (y0), n1 = lane_next (y1);
  if constexpr (HEDGE) {
    // index clamp of the reference: beyond the picture the neighbour is the edge sample
    const P lo = __builtin_amdgcn_perm (y0, y0, 0x01000100u), hi = __builtin_amdgcn_perm (y1, y1, 0x03020302u);
    p0 = is_first ? lo : p0;
    p1 = is_first ? lo : p1;
    n0 = is_last ? hi : n0;
    n1 = is_last ? hi : n1;
  }
  const P W[6] = { p0, p1, y0, y1, n0, n1 };
  P s[NT];
  gather_taps < F, K, 0, 0, NT > (W, s);
  const P x0 = lift_apply_pk < F, K > (st.target ? row[2] : row[0], s);
  gather_taps < F, K, 1, 0, NT > (W, s);
  const P x1 = lift_apply_pk < F, K > (st.target ? row[3] : row[1], s);
  if (st.target) {
    row[2] = x0;
    row[3] = x1;
  } else {
    row[0] = x0;
    row[1] = x1;
  }
}

template < int SH >
__device__ __forceinline__ P
out_round_pk (P x)
{
  if constexpr (SH == 1)
    return U ((S (x) + (short) 1) >> 1);        // orc_interleave2_rrshift1_s16: the add wraps
  else if constexpr (SH == 2)
    return U (S (x) - (S (x) >> 1));            // orc_haar_synth_rrshift1_int_s16: avgsw, no wrap
  else
    return x;
}

// all horizontal steps of one row, then round + interleave + one 16-byte store
// (COH bit 1: the row belongs to an intermediate LL image of the chain form)
// the combine form's per-row inputs: the prediction's 8 bytes for this lane (0x80 bytes: a picture without
// references) and how many of the lane's samples are inside the picture
struct Combine {
  u32x2 pred;
  bool row_ok;                  // the row is inside the picture (wave-uniform)
  bool full;                    // all 8 samples of the lane are: one 8-byte store
  bool any_ragged;              // (wave-uniform) some lane of the wave has 1 .. 7 samples inside
  int nvalid;                   // samples of the lane inside the picture (0 .. 8)
};

template < int F, bool HEDGE, int COH >
__device__ __forceinline__ void
finish_row (P (&row)[4], bool is_first, bool is_last, bool store_lane, char *dst, const Combine & cmb)
{
  hstep < F, 0, HEDGE > (row, is_first, is_last);
  hstep < F, 1, HEDGE > (row, is_first, is_last);
  if constexpr (filter_nsteps (F) == 4) {
    hstep < F, 2, HEDGE > (row, is_first, is_last);
    hstep < F, 3, HEDGE > (row, is_first, is_last);
  }
  constexpr int SH = filter_shift (F);
  const P a0 = out_round_pk < SH > (row[0]), a1 = out_round_pk < SH > (row[1]);
  const P b0 = out_round_pk < SH > (row[2]), b1 = out_round_pk < SH > (row[3]);
  u32x4 o;
  o.x = __builtin_amdgcn_perm (b0, a0, 0x05040100u);
  o.y = __builtin_amdgcn_perm (b0, a0, 0x07060302u);
  o.z = __builtin_amdgcn_perm (b1, a1, 0x05040100u);
  o.w = __builtin_amdgcn_perm (b1, a1, 0x07060302u);
  if constexpr ((COH & 4) != 0) {
    // r04 -- the picture, not the residual (see the header of this file): out = sat_u8 (residual + prediction) with the
    // reference's 16-bit wrapping add (orc_rrshift6_add_s16_2d: addw, convsuswb; the prediction is the (acc + 32) >> 6
    // of the OBMC launch, a u8 plane), or + 128 for a picture without references (orc_offsetconvert_u8_s16).
    // cmb.nvalid: how many of the lane's 8 samples of this row lie inside the picture.
    // r05: no per-lane branch and no loop in here.  r04 wrapped the epilogue in `if (lane stores)` and stored ragged lanes
    // in a loop over their valid bytes: 16 divergent regions with a waterfall loop each per tile, 25 k lines of code,
    // and every join a place where the prediction loads in flight were waited for.  Now every lane computes, a row
    // outside the picture is a SCALAR branch, the full lanes store under the exec mask, and the byte stores of a
    // ragged right edge (a picture width that is not a multiple of 8) sit behind a scalar test that is false for
    // every tile of such pictures as 2160p and 1080p.
    if (cmb.row_ok) {
      const uint32_t q[4] = { o.x, o.y, o.z, o.w };
      const uint32_t pw[4] = {
        __builtin_amdgcn_perm (0u, cmb.pred.x, 0x0c010c00u), __builtin_amdgcn_perm (0u, cmb.pred.x, 0x0c030c02u),
        __builtin_amdgcn_perm (0u, cmb.pred.y, 0x0c010c00u), __builtin_amdgcn_perm (0u, cmb.pred.y, 0x0c030c02u)
      };
      uint32_t v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        s16x2 t = S (q[k]) + S (pw[k]);
        t = __builtin_elementwise_min (__builtin_elementwise_max (t, (s16x2) (short) 0), (s16x2) (short) 255);
        v[k] = U (t);
      }
      u32x2 b;
      b.x = __builtin_amdgcn_perm (v[1], v[0], 0x06040200u);
      b.y = __builtin_amdgcn_perm (v[3], v[2], 0x06040200u);
      if (cmb.full) {
        gstore < u32x2 > (dst, b);       // (streaming stores here: a wash, HISTORY 8)
      }
      if (cmb.any_ragged) {
#pragma unroll
        for (int e = 0; e < 7; e++)
          if (!cmb.full && e < cmb.nvalid)
            gstore < uint8_t > (dst + e, (uint8_t) ((e < 4 ? b.x : b.y) >> (8 * (e & 3))));
      }
    }
  } else if (store_lane) {
    if constexpr ((COH & 2) != 0)
      coh_store16 (dst, o);
    else
      gstore < u32x4 > (dst, o);
  }
}

// LO..HI: region row pairs that exist in the picture (compile-time: see the header)
// COH (chain form): bit 0 the LL band is another tile's output of this launch, bit 1 so is this tile's output
struct NoWait {
  struct Seen { };
  __device__ __forceinline__ Seen ask () const { return Seen (); }
  __device__ __forceinline__ void settle (Seen) const { }
};

template < int F, int RP, int LO, int HI, bool HEDGE, int COH = 0, typename WAIT = NoWait >
__device__ __forceinline__ void
reg_tile (const IwtJob & job, int r0, int c0, int nr, int nc, int lane, WAIT wait = WAIT ())
{
  constexpr int H = filter_halo (F);
  P E[RP][4], O[RP][4];

  const int cl = c0 + 4 * lane;
  const uint32_t voff = (uint32_t) clampi (cl, 0, nc - 4) * 2u;
  constexpr Rng le = load_rows < F, RP, LO, HI > (0), lo = load_rows < F, RP, LO, HI > (1);
  // (chain form) ask how far the tiles that produce this tile's LL rows are, then load the detail bands -- they
  // come from the coefficient frame whatever the level above is doing --, then look at the answer (loads return
  // in order: it is there while the ~36 detail loads are still in flight), then load LL
  const auto seen = wait.ask ();
#pragma unroll
  for (int k = 0; k < RP; k++) {
    const int r = clampi (r0 + k, 0, nr - 1);
    if (k >= le.lo && k <= le.hi) {
      const u32x2 hl = IWT_LOAD8 ((const char *) job.sb[1] + (size_t) r * job.sb_stride[1] + voff);
      E[k][2] = hl.x;
      E[k][3] = hl.y;
    }
    if (k >= lo.lo && k <= lo.hi) {
      const u32x2 lh = IWT_LOAD8 ((const char *) job.sb[2] + (size_t) r * job.sb_stride[2] + voff);
      const u32x2 hh = IWT_LOAD8 ((const char *) job.sb[3] + (size_t) r * job.sb_stride[3] + voff);
      O[k][0] = lh.x;
      O[k][1] = lh.y;
      O[k][2] = hh.x;
      O[k][3] = hh.y;
    }
  }
  wait.settle (seen);
#pragma unroll
  for (int k = 0; k < RP; k++) {
    const int r = clampi (r0 + k, 0, nr - 1);
    if (k >= le.lo && k <= le.hi) {
      const u32x2 ll = IWT_LOAD8 ((const char *) job.sb[0] + (size_t) r * job.sb_stride[0] + voff);
      E[k][0] = ll.x;
      E[k][1] = ll.y;
    }
  }

  vstep < F, 0, RP, LO, HI > (E, O);
  vstep < F, 1, RP, LO, HI > (E, O);
  if constexpr (filter_nsteps (F) == 4) {
    vstep < F, 2, RP, LO, HI > (E, O);
    vstep < F, 3, RP, LO, HI > (E, O);
  }

  // lanes inside the picture: first / last substitute their edge sample for the neighbour
  const int l_lo = (max (0, -c0) + 3) >> 2, l_hi = min (63, ((nc - c0) >> 2) - 1);
  const bool is_first = lane == l_lo, is_last = lane == l_hi;
  const bool store_lane = lane >= max (l_lo, 1) && lane <= min (l_hi, 62);
  if constexpr ((COH & 4) != 0) {
    // combine form: the destination is the u8 picture (out_w x out_h inside the iwt-padded w x h).  The lane's 8
    // prediction bytes of a row are ONE aligned load (the host keeps rows of the prediction plane 8-byte aligned and
    // readable up to a multiple of 8 columns); a pair of rows is fetched while the pair before it is finished.
    const int x = 2 * cl, y0 = 2 * (r0 + H);
    const int nx = store_lane ? clampi (job.out_w - x, 0, 8) : 0;
    const bool full = nx == 8;
    const bool any_ragged = __builtin_amdgcn_readfirstlane ((int) (__ballot (nx > 0 && nx < 8) != 0)) != 0;
    // every lane gets an address it may read (and, if it stores at all, write): lanes outside the picture take column 0
    const int xa = nx > 0 ? x : 0;
    char *dst = (char *) job.dst + (size_t) y0 *job.dst_stride + xa;
    const bool has_pred = job.pred != nullptr;  // (uniform: a picture without references adds 128 instead)
    const uint8_t *pp = job.pred + xa;
    const u32x2 k128 = (u32x2) { 0x80808080u, 0x80808080u };
    // the lane's 8 prediction bytes of picture row y0 + 2 (i - H) + odd; rows below the picture read the last row
    // (computed, never stored): the load is unconditional, its row index a scalar
    auto fetch = [&](int i, int odd) {
      const int row = min (y0 + 2 * (i - H) + odd, job.out_h - 1);
      // (read once: a streaming load -- 8 x 2160p finest level 0.0801 -> 0.0790 ms; the picture as a streaming STORE
      // makes this launch 0.002 ms slower and the OBMC launches beside it 0.003 faster: left plain)
      return has_pred ? __builtin_nontemporal_load ((const SCHRO_GLOBAL u32x2 *) (pp + (size_t) row * job.pred_stride)) : k128;
    };
    // r05 -- where the prediction's registers come from.  The vertical steps have just finished: of the tile's RP row
    // pairs the H above and the H below the useful ones are dead now (16 H registers), and every finished pair gives
    // another 8 back; a pair's two prediction rows take 4.  So the predictions of the first AHEAD pairs are asked for
    // HERE, behind the vertical phase (a scheduling barrier keeps them from rising above it, where the whole tile is
    // live), and RAMP more pairs each time a pair is done, until all are on their way.  AHEAD 4 is what fits 128
    // registers without a spill for DD(9,7) (measured, 8 x 2160p finest level: AHEAD 1 / 2 / 3 / 4 = 0.0835 / 0.0787 /
    // 0.0768 / 0.0762 ms; 8 with 13 spilled dwords 0.0838 = r04's kernel, which also branched per lane, see finish_row).
    constexpr int NPAIR = RP - 2 * H;
    constexpr int AHEAD = cmin (NPAIR, 4), RAMP = 2;
    u32x2 pr[NPAIR][2];
    __builtin_amdgcn_sched_barrier (0);
#pragma unroll
    for (int j = 0; j < AHEAD; j++) {
      pr[j][0] = fetch (H + j, 0);
      pr[j][1] = fetch (H + j, 1);
    }
    __builtin_amdgcn_sched_barrier (0);
#pragma unroll
    for (int i = H; i < RP - H; i++) {
      const int j = i - H;
      const int y = y0 + 2 * j;
      const Combine c0 = { pr[j][0], y < job.out_h, full, any_ragged, nx }, c1 = { pr[j][1], y + 1 < job.out_h, full, any_ragged, nx };
      finish_row < F, HEDGE, COH > (E[i], is_first, is_last, store_lane, dst, c0);
      finish_row < F, HEDGE, COH > (O[i], is_first, is_last, store_lane, dst + job.dst_stride, c1);
      dst += 2 * (size_t) job.dst_stride;
      // pairs fetched so far: AHEAD + RAMP j; now RAMP more
#pragma unroll
      for (int t = 0; t < RAMP; t++) {
        const int nj = AHEAD + RAMP * j + t;
        if (nj < NPAIR) {
          pr[nj][0] = fetch (H + nj, 0);
          pr[nj][1] = fetch (H + nj, 1);
        }
      }
      // (no further ahead than that: left alone the scheduler hoists every row's prediction load to the top)
      __builtin_amdgcn_sched_barrier (0);
    }
  } else {
    char *dst = (char *) job.dst + (size_t) (2 * (r0 + H)) * job.dst_stride + (size_t) cl * 4;
    const Combine none = { (u32x2) { 0u, 0u }, false, false, false, 0 };
#pragma unroll
    for (int i = H; i < RP - H; i++) {
      finish_row < F, HEDGE, COH > (E[i], is_first, is_last, store_lane, dst, none);
      finish_row < F, HEDGE, COH > (O[i], is_first, is_last, store_lane, dst + job.dst_stride, none);
      dst += 2 * (size_t) job.dst_stride;
    }
  }
}

// tiles whose last N region row pairs lie below the picture
template < int F, int RP, int N, int COH, typename WAIT >
__device__ __forceinline__ void
reg_tile_bottom (int nout, const IwtJob & job, int r0, int c0, int nr, int nc, int lane, WAIT wait)
{
  if constexpr (N >= 1) {
    if (nout == N)
      reg_tile < F, RP, 0, RP - 1 - N, true, COH, WAIT > (job, r0, c0, nr, nc, lane, wait);
    else
      reg_tile_bottom < F, RP, N - 1, COH, WAIT > (nout, job, r0, c0, nr, nc, lane, wait);
  }
}

// Row placement of tile (tx, ty): it produces row pairs [ty UR, ty UR + UR); the last one is moved up
// to end at the picture's last row pair (it recomputes rows of the tile above: same values).  The top
// tile has its H halo row pairs above the picture; a tile near the bottom has nout = 0 .. H of its row
// pairs below it.  Both are compile-time cases.
template < int F, int RP >
__device__ __forceinline__ int
reg_tile_r0 (int ty, int nr)
{
  constexpr int H = filter_halo (F), UR = RP - 2 * H;
  int r0 = ty * UR - H;
  if (r0 + H + UR > nr)
    r0 = nr - UR - H;           // nr >= UR + H (host side)
  return r0;
}

template < int F, int RP, int COH, typename WAIT = NoWait >
__device__ __forceinline__ void
reg_tile_at (const IwtJob & job, int tx, int ty, int lane, WAIT wait = WAIT ())
{
  constexpr int H = filter_halo (F);
  const int nr = job.h >> 1, nc = job.w >> 1;
  const int r0 = reg_tile_r0 < F, RP > (ty, nr);
  const int nout = max (0, r0 + RP - nr);
  const int c0 = tx * kRegUC - 4;
  const bool hedge = c0 < 0 || c0 + 256 > nc;
  if (r0 < 0) {
    reg_tile < F, RP, H, RP - 1, true, COH, WAIT > (job, r0, c0, nr, nc, lane, wait);
  } else if (nout == 0) {
    if (!hedge)
      reg_tile < F, RP, 0, RP - 1, false, COH, WAIT > (job, r0, c0, nr, nc, lane, wait);
    else
      reg_tile < F, RP, 0, RP - 1, true, COH, WAIT > (job, r0, c0, nr, nc, lane, wait);
  } else {
    reg_tile_bottom < F, RP, H, COH, WAIT > (nout, job, r0, c0, nr, nc, lane, wait);
  }
}

// Register budget: four waves per SIMD (128 registers; DD(9,7) with 12 row pairs takes 106 -- left to itself the
// compiler spreads to 144, three waves).  Alone the kernel runs the same either way; beside the other batch's
// OBMC (seven waves of 69 registers on every SIMD) the smaller waves find room: finest level 0.0722 -> 0.0703 ms,
// the 8 x 2160p step 0.4054 -> 0.4012.  Five waves (102 registers) spill: 0.101 ms.
// (the Haar filters have no halo: all 12 row pairs are worked on, and 128 registers would spill 80 - 90 of them)
#define IIWT_REG_WAVES(F, RP) (((F) == 3 || (F) == 4) && (RP) == 12 ? 3 : 4)
// (the combine form of the 12-pair tiles: 4 waves spill ~40 registers of the epilogue and are still the faster
// launch -- 8 x 2160p finest level 0.081 ms against 0.086 ms at 3 waves without a spill, r04)
#define IIWT_REG_WAVES_M(F, RP, MODE) ((MODE) == 4 && (RP) == 12 ? 4 : IIWT_REG_WAVES (F, RP))
template < int F, int RP, int MODE >
__global__ __launch_bounds__ (kRegThreads) __attribute__ ((amdgpu_waves_per_eu (IIWT_REG_WAVES_M (F, RP, MODE), IIWT_REG_WAVES_M (F, RP, MODE))))
void iiwt_reg_kernel (const IwtJob * __restrict__ jobs, int njobs, int total_tiles)
{
  const int wg = xcd_tile_id (blockIdx.x, gridDim.x);
  // the wave index is uniform, but only readfirstlane tells the compiler: with it the tile
  // origin, row addresses and edge tests live in SGPRs and branch on SCC
  const int tile = __builtin_amdgcn_readfirstlane (wg * (kRegThreads / 64) + (threadIdx.x >> 6));
  if (tile >= total_tiles)
    return;
  const int lane = threadIdx.x & 63;
  const IwtJob job = jobs[find_job (jobs, njobs, tile)];
  const int t = tile - job.tile_base;
  reg_tile_at < F, RP, MODE > (job, t % job.tiles_x, t / job.tiles_x, lane);
}

// rows per wave of the small form: 4 useful row pairs whatever the halo
constexpr int
small_rp (int f)
{
  return 4 + 2 * filter_halo (f);
}

// ---- r04: the chain form -- every level of every plane of a batch in ONE launch ------------------------
// A launch per level runs the levels one after the other: the coarse ones (25 + 100 MB of a 523 MB transform
// of 8 x 2160p) are two launches of their own whose waves all start together, load, compute and store in
// phases, and leave the chip idle between them (8 x 1080p: 21.8 + 7.8 + 7.8 us, 33 % of the HBM roofline
// where the finest level alone runs at 57 %).  Here a tile of level l starts as soon as the tiles of level
// l + 1 that produce its LL rows have finished:
//   * the host hands out the tiles of all levels in ONE order (iiwt_chain_order): sorted so that a producer
//     always precedes its consumers and follows them closely (coarse rows run just ahead of the finer rows
//     they feed), plane by plane to the XCDs;
//   * workgroup b takes tiles 4 b .. 4 b + 3 of that order.  The order is topological and the dispatcher starts
//     the workgroups of an XCD in index order, so the lowest unfinished workgroup always runs and waits for
//     nobody: no deadlock.  (A ticket counter -- "the next four tiles" by an atomic, which needs no assumption
//     about the dispatcher -- was built first: 4.6 k atomics on ONE address per 8 x 2160p launch serialise,
//     0.177 ms against 0.139 without; the same for a counter of finished workgroups.)
//   * a finished tile of a level > 0 counts itself in its (job, tile row) counter; a tile with a producer
//     polls the one to three counters of the producer rows it reads -- bounded: after 2^22 polls it gives up,
//     writes the launch's epoch to a pinned host word (the next call reports it) and runs on;
//   * the counters are never reset: launch number n of a geometry on a queue waits for n times the row's tiles
//     (the host zeroes them when the geometry changes or a launch gave up);
//   * the intermediate LL images are written through (coh_store16 above) and read with ordinary loads.
constexpr int kChainPolls = 1 << 22;

// the wait of a tile for the producer tile rows it reads: called by reg_tile between its detail-band loads and
// its LL loads.  The counters are asked for together; a tile whose producers are long done (every tile of the
// finest level, in the level-major order) pays one round trip that overlaps the loads in flight.
struct ChainWait {
  const uint32_t *c;
  int n;
  uint32_t want;
  uint32_t *gave_up;
  uint32_t epoch;
  struct Seen {
    uint32_t v0, v1, v2;
  };
  __device__ __forceinline__ Seen ask () const
  {
    Seen s = { want, want, want };
    if (n > 0)
      s.v0 = __hip_atomic_load (c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n > 1)
      s.v1 = __hip_atomic_load (c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n > 2)
      s.v2 = __hip_atomic_load (c + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return s;
  }
  __device__ __forceinline__ void settle (Seen s) const
  {
    int polls = 0;
    while (min (s.v0, min (s.v1, s.v2)) < want) {
      __builtin_amdgcn_s_sleep (2);
      if (++polls > kChainPolls) {
        if ((threadIdx.x & 63) == 0)
          __hip_atomic_store (gave_up, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      s = ask ();
    }
    asm volatile ("":::"memory");       // (the LL loads stay behind the wait)
  }
};

template < int F >
__global__ __launch_bounds__ (kRegThreads) __attribute__ ((amdgpu_waves_per_eu (IIWT_REG_WAVES (F, kRegRP), IIWT_REG_WAVES (F, kRegRP))))
void iiwt_chain_kernel (const IwtJob * __restrict__ jobs, const uint32_t * __restrict__ order, int n_tiles,
    uint32_t * __restrict__ ctrl, uint32_t run, uint32_t * __restrict__ gave_up, uint32_t epoch)
{
  const int lane = threadIdx.x & 63;
  const int pos = __builtin_amdgcn_readfirstlane ((int) blockIdx.x * (kRegThreads / 64) + (int) (threadIdx.x >> 6));
  if (pos >= n_tiles)
    return;
  const uint32_t entry = __builtin_amdgcn_readfirstlane (gload < uint32_t > (order + pos));
  const IwtJob job = jobs[entry >> 16];
  const int t = (int) (entry & 0xffffu);
  const int tx = t % job.tiles_x, ty = t / job.tiles_x;
  // the producer rows this tile reads: its LL rows = output rows of the producer job, whose tile row r covers
  // rows [r dep_rows2, (r + 1) dep_rows2) (its last one, moved up, the rest): one to three counters
  ChainWait wait;
  wait.n = 0;
  if (job.dep_rows2 > 0) {
    const int nr = job.h >> 1;
    const int r0 = job.small ? reg_tile_r0 < F, small_rp (F) > (ty, nr) : reg_tile_r0 < F, kRegRP > (ty, nr);
    const int rp = job.small ? small_rp (F) : kRegRP;
    const int a = max (r0, 0), b = min (r0 + rp, nr);
    const int lo = min (a / job.dep_rows2, job.dep_tiles_y - 1), hi = min ((b - 1) / job.dep_rows2, job.dep_tiles_y - 1);
    wait.c = ctrl + job.dep_ctr + lo;
    wait.n = min (hi - lo + 1, 3);      // (12 LL rows over tile rows of >= 8)
    wait.want = run * (uint32_t) job.dep_tiles_x;
    wait.gave_up = gave_up;
    wait.epoch = epoch;
  }
  if (job.ctr >= 0) {           // an intermediate level: its output is read by tiles of this launch
    if (job.small)
      reg_tile_at < F, small_rp (F), 2, ChainWait > (job, tx, ty, lane, wait);
    else
      reg_tile_at < F, kRegRP, 2, ChainWait > (job, tx, ty, lane, wait);
    // every store of the wave has gone through before the tile counts as done
    __builtin_amdgcn_s_waitcnt (0x0f70);        // vmcnt (0)
    if (lane == 0)
      __hip_atomic_fetch_add (&ctrl[job.ctr + ty], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (job.small)
      reg_tile_at < F, small_rp (F), 0, ChainWait > (job, tx, ty, lane, wait);
    else
      reg_tile_at < F, kRegRP, 0, ChainWait > (job, tx, ty, lane, wait);
  }
}

template < int F >
int
launch_chain (hipStream_t stream, const IwtJob * d_jobs, const uint32_t * d_order, int n_tiles, uint32_t * ctrl, uint32_t run,
    uint32_t * gave_up, uint32_t epoch)
{
  const int wgs = (n_tiles + kRegThreads / 64 - 1) / (kRegThreads / 64);
  SCHRO_LAUNCH ((iiwt_chain_kernel < F >), dim3 (wgs), dim3 (kRegThreads), 0, stream, d_jobs, d_order, n_tiles, ctrl, run,
      gave_up, epoch);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "iiwt (chain form) launch: %s", hipGetErrorString (e));
  return 0;
}

template < int F >
int
launch_reg (hipStream_t stream, const IwtJob * d_jobs, int njobs, int total_tiles, bool small, bool combine)
{
  const int wgs = (total_tiles + kRegThreads / 64 - 1) / (kRegThreads / 64);
  if (small && combine)
    SCHRO_LAUNCH ((iiwt_reg_kernel < F, small_rp (F), 4 >), dim3 (wgs), dim3 (kRegThreads), 0, stream,
        d_jobs, njobs, total_tiles);
  else if (combine)
    SCHRO_LAUNCH ((iiwt_reg_kernel < F, kRegRP, 4 >), dim3 (wgs), dim3 (kRegThreads), 0, stream, d_jobs,
        njobs, total_tiles);
  else if (small)
    SCHRO_LAUNCH ((iiwt_reg_kernel < F, small_rp (F), 0 >), dim3 (wgs), dim3 (kRegThreads), 0, stream,
        d_jobs, njobs, total_tiles);
  else
    SCHRO_LAUNCH ((iiwt_reg_kernel < F, kRegRP, 0 >), dim3 (wgs), dim3 (kRegThreads), 0, stream, d_jobs,
        njobs, total_tiles);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "iiwt (register form) launch: %s", hipGetErrorString (e));
  return 0;
}

}                               // namespace

// which (filter, sample size) the register form is built for; tile geometry
bool
iiwt_reg_supported (int filter, int bpp)
{
  return bpp == 2 && filter != 5 && filter >= 0 && filter <= 6;
}

// Tile geometry of the large form (12 row pairs per wave: least halo re-reading, for levels
// that are bandwidth) or the small one (4 useful row pairs per wave: levels of a few hundred
// tiles are the latency of one wave's serial work, so halve it and double the waves).
// A plane needs at least min_row_pairs sub-band rows (top and bottom edge in different tiles).
void
iiwt_reg_geometry (int filter, int small, int *useful_cols, int *useful_row_pairs, int *min_row_pairs)
{
  const int rp = small ? small_rp (filter) : kRegRP;
  *useful_cols = kRegUC;
  *useful_row_pairs = rp - 2 * filter_halo (filter);
  *min_row_pairs = rp - filter_halo (filter);
}

// (the chain form is a measured-slower form: built into the experiments library only, schro_hip_internal.h)
int
launch_iiwt_chain (hipStream_t stream, const IwtJob * d_jobs, const uint32_t * d_order, int n_tiles, uint32_t * ctrl,
    uint32_t run, uint32_t * gave_up, uint32_t epoch, int filter)
{
#ifdef SCHRO_HIP_EXPERIMENTS
  switch (filter) {
    case 0: return launch_chain < 0 > (stream, d_jobs, d_order, n_tiles, ctrl, run, gave_up, epoch);
    case 1: return launch_chain < 1 > (stream, d_jobs, d_order, n_tiles, ctrl, run, gave_up, epoch);
    case 2: return launch_chain < 2 > (stream, d_jobs, d_order, n_tiles, ctrl, run, gave_up, epoch);
    case 3: return launch_chain < 3 > (stream, d_jobs, d_order, n_tiles, ctrl, run, gave_up, epoch);
    case 4: return launch_chain < 4 > (stream, d_jobs, d_order, n_tiles, ctrl, run, gave_up, epoch);
    case 6: return launch_chain < 6 > (stream, d_jobs, d_order, n_tiles, ctrl, run, gave_up, epoch);
  }
  return set_error (SCHRO_HIP_EINVAL, "iiwt (chain form): filter %d not built", filter);
#else
  (void) stream, (void) d_jobs, (void) d_order, (void) n_tiles, (void) ctrl, (void) run, (void) gave_up, (void) epoch, (void) filter;
  return set_error (SCHRO_HIP_EUNSUPPORTED, "the chain form of the register wavelet is built into the experiments library only");
#endif
}

int
launch_iiwt_reg (hipStream_t stream, const IwtJob * d_jobs, int njobs, int total_tiles, int filter,
    int small, int combine)
{
  switch (filter) {
    case 0: return launch_reg < 0 > (stream, d_jobs, njobs, total_tiles, small, combine);
    case 1: return launch_reg < 1 > (stream, d_jobs, njobs, total_tiles, small, combine);
    case 2: return launch_reg < 2 > (stream, d_jobs, njobs, total_tiles, small, combine);
    case 3: return launch_reg < 3 > (stream, d_jobs, njobs, total_tiles, small, combine);
    case 4: return launch_reg < 4 > (stream, d_jobs, njobs, total_tiles, small, combine);
    case 6: return launch_reg < 6 > (stream, d_jobs, njobs, total_tiles, small, combine);
  }
  return set_error (SCHRO_HIP_EINVAL, "iiwt (register form): filter %d not built", filter);
}

}                               // namespace schro
