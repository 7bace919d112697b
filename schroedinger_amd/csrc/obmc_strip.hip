// obmc_strip.hip -- r05: OBMC for the headline's block geometry with the accumulator in REGISTERS.
//
// What it computes: schro_motion_render_u8 (schromotion8.c:700-929; block arithmetic :542-657, get_block :303-335,
// the quarter-pel fetch schroframe.c:2288-2482, the weights schromotion.c:40-83) for one plane with default picture
// weights, half- or quarter-pel references, 12 x 12 blocks every 8 pixels (the 12/8 block set: luma of BASELINE
// config 3) -- the same bytes as obmc_row.hip, by another route.
//
// Why a second form.  obmc_row.hip scatters (block, row) items into an LDS accumulator tile: per 128 x 32 tile a
// workgroup decodes ~100 blocks, sorts their rows into classes, runs ~15 passes and reads the tile back.  Its
// counters (profiles/r04_pmc_summary.txt, r05 mix) say where the time goes: 738 vector and 475 scalar instructions
// per wave of which barely half are the passes; every unit (vector, scalar, LDS, texture path) is 40 - 55 % busy and
// the waves sit in s_waitcnt / s_barrier 60 % of their lives.  Two things the block grid offers make all of that
// unnecessary:
//   * blocks only overlap their direct neighbours (12 <= 2 x 8), and the overlap is the same 4 pixels everywhere:
//     with a lane group per block COLUMN the horizontal sum is one cross-lane exchange of two registers, and with a
//     wave walking DOWN its columns the vertical sum is a carry of four registers from one block row to the next;
//   * the weights are separable and their folded forms at the picture's rim are the constant 8
//     (schromotion8.c:673-693: the ramps of two neighbours add up to 8), so a lane keeps its x weights in six
//     registers for the whole launch and the y weight is one multiplier per row.
// So: a wave owns 16 block columns (the first is the halo: the left neighbour of its first useful column) and
// walks a segment of block rows; lane 4 c + q predicts row 4 g + q of block column c in step g = 0, 1, 2 of a block
// (four consecutive rows of a window in four consecutive lanes: the 128-byte lines of the tiled half-pel planes
// hold four rows, so the texture path sees the same line sharing as in the row kernel); products p * wx, the
// exchange with the lane four to the left, * wy, the carry, rounding, and the 8 finished pixels of the row go out.
// No LDS tile, no barrier, no sort, no item list, no class loops: every lane runs the same instructions all the
// time -- the both-references, four-tap path, whatever the block's mode (a block that does not use a reference or
// a tap reads its first tap again: lerp (a, a) = a; the mode selects afterwards).  DC blocks ride along as 16-bit
// values, so DC values outside 8 bits wrap exactly as the reference's s16 arithmetic does (no exact-add path).
// Windows that leave the reference vertically are clamped row by row in every step (fetch's CLAMP on y); columns
// need no clamp (aprons, schro_hip_internal.h).
//
// Cost per step of a lane (one block row of 12 pixels): 8 buffer loads, ~165 vector instructions, two
// ds_bpermute; against the row kernel's 247 vector + 160 scalar instructions per block row all told.
//
// Geometry handled here: xblen = yblen = 12, xbsep = ybsep = 8 (offsets 2), one byte per sample; everything else
// stays with obmc_row.hip / obmc.hip (plane_obmc.cpp decides).
//
// MEASURED (r05, 8 x 2160p luma, prediction only): bit-exact on every test of tests/test_gpu_obmc.py / _combine / _fuzz /
// _stream -- and 2.5 x SLOWER than the row kernel: 0.25 - 0.27 ms per launch against 0.105 (one item per wave, 8 waves per
// SIMD, 8976 waves on 8192 slots: 0.286; items dealt statically, both references' loads in flight, the next block's
// vectors prefetched: 0.25 at 5 waves per SIMD, 0.27 at 6).  Why: a step covers FOUR rows of a window -- 1.75 lines of 128
// bytes per tap plane where the row kernel's twelve consecutive lanes touch 3.75 per twelve rows (x 1.4) --, every lane
// loads all eight taps (x 2.7 against the 2.9 taps a block needs on average; a tap a block does not use repeats the first
// tap's address, but the repeat is issued before the first has landed and goes to L2 as well), and a wave's step reads
// ~170 lines = 21 KB: with 20 - 32 waves per CU nothing survives in the 32 KB L1.  The launch moves ~3 x the row kernel's
// lines from L2 and every step is a full memory round trip with ~165 vector instructions behind it.  (The row kernel itself is
// NOT bound by its lines: with every window on one tap its luma launch takes 0.088 ms whether the windows share lines or
// not, HISTORY.md section 8.)  With the unused taps masked off this form would still move 1.4 x the lines.  It stays in the
// experiments build (SCHRO_HIP_OBMC_STRIP=1) as the third formulation the parity tests can compare.

#include "schro_hip_internal.h"
#include "obmc_common.h"

namespace schro {
#ifdef SCHRO_HIP_EXPERIMENTS
namespace {

constexpr int kSThreads = 256;
constexpr int kSCols = 16;              // block columns per wave: column 0 is the halo
constexpr int kSUseful = kSCols - 1;

typedef unsigned short u16x2 __attribute__ ((ext_vector_type (2)));
typedef short s16x2 __attribute__ ((ext_vector_type (2)));

__device__ __forceinline__ uint32_t
pk_mul (uint32_t a, uint32_t b)
{
  return __builtin_bit_cast (uint32_t, (u16x2) (__builtin_bit_cast (u16x2, a) * __builtin_bit_cast (u16x2, b)));
}

__device__ __forceinline__ uint32_t
pk_add (uint32_t a, uint32_t b)
{
  return __builtin_bit_cast (uint32_t, (u16x2) (__builtin_bit_cast (u16x2, a) + __builtin_bit_cast (u16x2, b)));
}

__device__ __forceinline__ uint32_t
lerp1 (uint32_t a, uint32_t b)
{
  return __builtin_amdgcn_lerp (a, b, 0x01010101u);     // per byte (a + b + 1) >> 1 = avgub
}

// obmc_weight_1d (schromotion.c:57-69) for the 12 / 8 geometry (offset 2: the ramp 1 3 5 7)
__device__ __forceinline__ int
weight12 (int i, int blen, int offset)
{
  if (offset == 0)
    return 8;
  int x = i;
  if (i >= 2 * offset) {
    if (blen - 1 - i >= 2 * offset)
      return 8;
    x = blen - 1 - i;
  }
  if (offset == 1)
    return x == 0 ? 3 : 5;
  return 1 + (6 * x + offset - 1) / (2 * offset - 1);
}

// one reference's window of the lane's block, as every step addresses it
struct StripRef {
  uint32_t colbase;             // chunk * 512 + byte in the chunk + 128 if the first column is an h-half one (0: not used)
  int hy;                       // half-pel row of the block's first sample row (signed)
  int dB;                       // the X + 1 taps are dB bytes on (0: not used)
  int ry;                       // 1: the window sits at a vertical quarter position (taps of rows Y and Y + 1)
};

// byte offset of plane row y: band (4 rows) * stride + 32 * row in the band
__device__ __forceinline__ uint32_t
row_ofs (uint32_t y, uint32_t stride)
{
  return __umul24 (y >> 2, stride) + ((y & 3u) << 5);
}

// the 12 prediction bytes of sample row `row` of the window: the four-tap form whatever the phase (a tap the phase does
// not use is the first tap again).  Every sample row is clamped to the image on its own (fetch's CLAMP on y).
// Two halves: the four loads of a reference go out (both references' eight before the first result is touched: one
// memory round trip per step), then the bytes are shifted into place and averaged.
struct StripTaps {
  u32x4 q[4];
  uint32_t sh;                  // the four taps' byte shifts, 2 bits each
};

__device__ __forceinline__ void
strip_issue (__amdgpu_buffer_rsrc_t ref, uint32_t stride, int gh, const StripRef & rr, int row, StripTaps & t)
{
  const int hyr = rr.hy + 2 * row;
  const uint32_t Y0 = (uint32_t) clampi (hyr, 0, gh), Y1 = (uint32_t) clampi (hyr + rr.ry, 0, gh);
  const uint32_t offA = rr.colbase + ((Y0 & 1u) << 8) + row_ofs (Y0 >> 1, stride);
  const uint32_t offC = rr.colbase + ((Y1 & 1u) << 8) + row_ofs (Y1 >> 1, stride);
  const uint32_t offB = offA + (uint32_t) rr.dB, offD = offC + (uint32_t) rr.dB;
  t.q[0] = __builtin_amdgcn_raw_buffer_load_b128 (ref, (int) (offA & ~3u), 0, 0);
  t.q[1] = __builtin_amdgcn_raw_buffer_load_b128 (ref, (int) (offB & ~3u), 0, 0);
  t.q[2] = __builtin_amdgcn_raw_buffer_load_b128 (ref, (int) (offC & ~3u), 0, 0);
  t.q[3] = __builtin_amdgcn_raw_buffer_load_b128 (ref, (int) (offD & ~3u), 0, 0);
  t.sh = (offA & 3u) | ((offB & 3u) << 2) | ((offC & 3u) << 4) | ((offD & 3u) << 6);
}

__device__ __forceinline__ void
strip_finish (const StripTaps & t, uint32_t * out)
{
  const uint32_t sa = t.sh & 3u, sb = (t.sh >> 2) & 3u, sc = (t.sh >> 4) & 3u, sd = t.sh >> 6;
  const uint32_t ra[4] = { t.q[0].x, t.q[0].y, t.q[0].z, t.q[0].w }, rb[4] = { t.q[1].x, t.q[1].y, t.q[1].z, t.q[1].w };
  const uint32_t rc[4] = { t.q[2].x, t.q[2].y, t.q[2].z, t.q[2].w }, rd[4] = { t.q[3].x, t.q[3].y, t.q[3].z, t.q[3].w };
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const uint32_t a = __builtin_amdgcn_alignbyte (ra[k + 1], ra[k], sa), b = __builtin_amdgcn_alignbyte (rb[k + 1], rb[k], sb);
    const uint32_t c = __builtin_amdgcn_alignbyte (rc[k + 1], rc[k], sc), d = __builtin_amdgcn_alignbyte (rd[k + 1], rd[k], sd);
    // per byte (a + b + c + d + 2) >> 2 exactly (obmc_row.hip: predict_row); a + a + c + c, a + b + a + b, 4 a fall out of it
    const uint32_t h0 = lerp1 (a, b), h1 = lerp1 (c, d);
    out[k] = __builtin_amdgcn_lerp (h0, h1, ~((a ^ b) | (c ^ d)));
  }
}

#define SCHRO_STRIP_WAVES 6
// Work is handed out statically: the launch has at most as many waves as the device holds at once, wave w takes items w,
// w + waves, ... (an item = a strip of a segment of a plane) -- every wave the same number of them, give or take one: a
// launch of 1.1 "rounds" of one item per wave took twice a wave's life.
template < bool NORES >
__global__ __launch_bounds__ (kSThreads) __attribute__ ((amdgpu_waves_per_eu (SCHRO_STRIP_WAVES, SCHRO_STRIP_WAVES)))
void obmc_strip_kernel (const ObmcJob * __restrict__ jobs, int njobs, int total_items, int total_waves, int seg_rows, uint32_t * __restrict__ overflow)
{
  const int wave0 = __builtin_amdgcn_readfirstlane ((int) (blockIdx.x * (kSThreads / 64) + (threadIdx.x >> 6)));
  for (int item = wave0; item < total_items; item += total_waves) {
  const ObmcJob job = jobs[find_job (jobs, njobs, item)];
  const int t = item - job.tile_base;
  const int seg = t / job.tiles_x, strip = t - seg * job.tiles_x;       // (tiles_x: strips per segment)
  const int lane = threadIdx.x & 63, q = lane & 3, cl = lane >> 2;
  const int i = strip * kSUseful - 1 + cl;      // the lane's block column; -1 and >= nbx: no block
  const bool have = i >= 0 && i < job.nbx;
  const int prec = job.prec, gh = 2 * job.h - 2;
  constexpr int yblen = 12, ybsep = 8, xbsep = 8, xoff = 2, yoff = 2;     // (obmc_strip_ok)

  // the lane's x weights, two pixels per word, folded at the picture's left / right rim; zero where there is no block
  uint32_t wx[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    int w0 = weight12 (2 * k, 12, xoff), w1 = weight12 (2 * k + 1, 12, xoff);
    if (i == 0 && 2 * k + 1 < 2 * xoff) {       // accumulate_slow's folding (schromotion8.c:673-693), by edge type
      w0 += weight12 (2 * xoff - 2 * k - 1, 12, xoff);
      w1 += weight12 (2 * xoff - 2 * k - 2, 12, xoff);
    }
    if (i == job.nbx - 1 && 2 * k >= xbsep) {
      w0 += weight12 (2 * (12 - xoff) - 2 * k - 1, 12, xoff);
      w1 += weight12 (2 * (12 - xoff) - 2 * k - 2, 12, xoff);
    }
    wx[k] = have ? (uint32_t) w0 | ((uint32_t) w1 << 16) : 0u;
  }
  // ... and its y weights for the rows it takes in steps 0 and 2 (step 1: rows 4 .. 7 of 12, the flat part)
  const uint32_t wy0 = (uint32_t) weight12 (q, yblen, yoff) * 0x00010001u, wy1 = (uint32_t) weight12 (4 + q, yblen, yoff) * 0x00010001u;
  const uint32_t wy2 = (uint32_t) weight12 (8 + q, yblen, yoff) * 0x00010001u;
  const uint32_t wy0_top = (uint32_t) (weight12 (q, yblen, yoff) + (q < 2 * yoff ? weight12 (2 * yoff - q - 1, yblen, yoff) : 0)) * 0x00010001u;
  const uint32_t wy2_bot = (uint32_t) (weight12 (8 + q, yblen, yoff) + weight12 (2 * (yblen - yoff) - (8 + q) - 1, yblen, yoff)) * 0x00010001u;

  __amdgpu_buffer_rsrc_t rsrc[2];
  uint32_t rstride[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    rstride[r] = (uint32_t) job.ref_stride[r];
    rsrc[r] = __builtin_amdgcn_make_buffer_rsrc ((void *) job.ref[r], 0, (int) ((uint32_t) job.ref_stride[r] * (uint32_t) ((job.h + 3) >> 2)), 0x00020000);
  }

  // block rows of this segment; the one in front of it only for its last four rows (what it carries into the segment)
  const int nby_touch = min (job.nby, (job.h + yoff + ybsep - 1) / ybsep);
  const int j_lo = seg * seg_rows, j_hi = min (j_lo + seg_rows, nby_touch);
  const int x0 = xbsep * i - xoff;      // the lane group's 8 output pixels start here
  // (column nbx is no block, but its lane group receives the last block's pixels 8 .. 11: the picture's last columns)
  const bool store_lane = cl >= 1 && i >= 0 && i <= job.nbx;
  const int bperm_addr = ((lane - 4) & 63) << 2;
  uint32_t carry[4] = { 0u, 0u, 0u, 0u };

  // the motion vector record of the block one row on is asked for while this row's block is worked on
  auto mv_fetch = [&](int jj, uint32_t * rec) {
    rec[0] = rec[1] = rec[2] = 0u;
    if (have && jj < j_hi) {
      const uint8_t *mvp = job.mvs + (size_t) 20 * ((size_t) jj * job.nbx + i);
      rec[0] = gload < uint32_t > (mvp);
      rec[1] = gload < uint32_t > (mvp + 12);
      rec[2] = gload < uint32_t > (mvp + 16);
    }
  };
  uint32_t mv_next[3];
  mv_fetch (max (j_lo - 1, 0), mv_next);
  for (int jj = max (j_lo - 1, 0); jj < j_hi; jj++) {
    // ---- the lane's block (i, jj): every lane of the group decodes it (the four read the same 20 bytes) ----
    const int by = ybsep * jj - yoff;
    uint32_t mode = 0u, dcw = 0u;
    bool wide = false;
    StripRef rr[2] = { {0u, 0, 0, 0}, {0u, 0, 0, 0} };
    const uint32_t flags = mv_next[0], v01 = mv_next[1], v23 = mv_next[2];
    mv_fetch (jj + 1, mv_next);
    if (have) {
      mode = flags & 3u;
      const bool interior = i >= 1 && i < job.max_x_blocks && jj >= 1 && jj < job.max_y_blocks;
      const int dcv = job.comp == 0 ? (int16_t) (v01 & 0xffff) : job.comp == 1 ? (int16_t) (v01 >> 16) : (int16_t) (v23 & 0xffff);
      // get_dc_block stores a uint8_t; block_acc_dc multiplies a 16-bit parameter
      const int pdc = interior ? (int) (int16_t) (dcv + 128) : (int) (uint8_t) (dcv + 128);
      dcw = ((uint32_t) pdc & 0xffffu) * 0x00010001u;
      wide = mode == 0u && (unsigned) pdc > 255u;
      const int bx = xbsep * i - xoff;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (!(mode & (uint32_t) (r + 1)))
          continue;
        int fx, fy;
        mv_origin (job, bx, by, v01, v23, r, &fx, &fy);
        const int hx = prec >= 2 ? fx >> 1 : fx, hy = prec >= 2 ? fy >> 1 : fy;
        const int rx = prec >= 2 ? fx & 1 : 0, ry = prec >= 2 ? fy & 1 : 0;
        const int xp = (hx >> 1) + kHpApron, px = hx & 1;
        rr[r].colbase = (uint32_t) ((xp >> 4) * 512 + (xp & 15) + px * 128);
        rr[r].hy = hy;
        rr[r].dB = rx ? (px ? 1 - 128 : 128) : 0;
        rr[r].ry = ry;
      }
    }
    if (NORES && overflow && wide)      // (prediction_only launches: such a prediction does not fit the u8 plane it is written to)
      __hip_atomic_store (overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const bool any_wide = __ballot (wide) != 0;
    const bool top = jj == 0, bottom = jj == job.nby - 1;
    // (the block row in front of the segment: its last step only)
#pragma unroll 1
    for (int g = jj < j_lo ? 2 : 0; g < 3; g++) {
      const int row = 4 * g + q;
      uint32_t p0[3], p1[3];
      {
        StripTaps t0, t1;
        strip_issue (rsrc[0], rstride[0], gh, rr[0], row, t0);
        strip_issue (rsrc[1], rstride[1], gh, rr[1], row, t1);
        strip_finish (t0, p0);
        strip_finish (t1, p1);
      }
      uint32_t tt[6];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        // one reference: averaged with itself; none (DC): below
        const uint32_t a = (mode & 1u) ? p0[k] : p1[k], b = (mode & 2u) ? p1[k] : p0[k];
        const uint32_t p = lerp1 (a, b);        // avgub of the two predictions, schromotion8.c:560-566 with the default weights
        tt[2 * k] = __builtin_amdgcn_perm (0u, p, 0x0c010c00u);
        tt[2 * k + 1] = __builtin_amdgcn_perm (0u, p, 0x0c030c02u);
      }
      if (mode == 0u) {
#pragma unroll
        for (int k = 0; k < 6; k++)
          tt[k] = dcw;
      }
#pragma unroll
      for (int k = 0; k < 6; k++)
        tt[k] = pk_mul (tt[k], wx[k]);
      // the four pixels this block shares with its left neighbour: that block's pixels 8 .. 11 of the same row
      const uint32_t n4 = (uint32_t) __builtin_amdgcn_ds_bpermute (bperm_addr, (int) tt[4]);
      const uint32_t n5 = (uint32_t) __builtin_amdgcn_ds_bpermute (bperm_addr, (int) tt[5]);
      const uint32_t wy = g == 0 ? (top ? wy0_top : wy0) : g == 1 ? wy1 : (bottom ? wy2_bot : wy2);
      uint32_t o[4];
      o[0] = pk_mul (pk_add (tt[0], n4), wy);
      o[1] = pk_mul (pk_add (tt[1], n5), wy);
      o[2] = pk_mul (tt[2], wy);
      o[3] = pk_mul (tt[3], wy);
      if (g == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++)
          o[k] = pk_add (o[k], carry[k]);
      }
      if (g == 2 && !bottom) {
        // rows 8 .. 11: the next block row's rows 0 .. 3 complete them
#pragma unroll
        for (int k = 0; k < 4; k++)
          carry[k] = o[k];
        continue;
      }
      const int y = by + row;
      if (y < 0 || y >= job.h || !store_lane)
        continue;
      uint32_t v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        s16x2 s = (__builtin_bit_cast (s16x2, o[k]) + (short) 32) >> 6;         // orc_rrshift6_add_s16_2d, schroorc.orc:636-661
        if constexpr (!NORES) {
          if (job.residual && x0 >= 0 && x0 + 8 <= job.w) {
            const uint32_t rw = gload < u32_u > ((const char *) job.residual + (size_t) y * job.residual_stride + 2 * (x0 + 2 * k));
            s = s + __builtin_bit_cast (s16x2, rw);
          }
        }
        s = __builtin_elementwise_min (__builtin_elementwise_max (s, (s16x2) (short) 0), (s16x2) (short) 255);
        v[k] = __builtin_bit_cast (uint32_t, s);
      }
      uint8_t *d = job.out + (size_t) y * job.out_stride + x0;
      if (x0 >= 0 && x0 + 8 <= job.w) {
        u32x2 b;
        b.x = __builtin_amdgcn_perm (v[1], v[0], 0x06040200u);
        b.y = __builtin_amdgcn_perm (v[3], v[2], 0x06040200u);
        gstore < u32x2_u > (d, b);       // (the run starts 2 pixels in front of a multiple of 8)
      } else {
        // the picture's left / right rim: pixel by pixel (with the residual, where there is one)
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int x = x0 + e;
          if (x < 0 || x >= job.w)
            continue;
          int val = (int) (int16_t) ((o[e >> 1] >> (16 * (e & 1))) & 0xffffu);
          val = (int) (int16_t) (val + 32) >> 6;
          if constexpr (!NORES) {
            if (job.residual)
              val = (int) (int16_t) (val + (int) gload < int16_t > ((const int16_t *) ((const char *) job.residual + (size_t) y * job.residual_stride) + x));
          }
          gstore < uint8_t > (d + e, (uint8_t) clampi (val, 0, 255));
        }
      }
    }
    (void) any_wide;
  }
  }                             // (items)
}

}                               // namespace
#endif

// which planes the strip kernel takes: the 12 / 8 block set on a one-byte-per-sample plane, default weights (the caller
// has checked those: it is a row-kernel job), an s16 residual or none
bool
obmc_strip_ok (const ObmcJob & j)
{
  return j.xblen == 12 && j.yblen == 12 && j.xbsep == 8 && j.ybsep == 8 && j.xoff == 2 && j.yoff == 2 && j.ref_ps == 0
      && (j.prec == 1 || j.prec == 2) && (!j.residual || j.res_bpp == 2) && j.nbx >= 2 && j.nby >= 2 && !j.out_s16
      && j.w1 == 1 && j.w2 == 1 && j.wbits == 1;        // (the default weights: the row kernels' fades, r06, are not its case)
}

// waves of one plane: strips of 15 block columns x segments of seg_rows block rows
void
obmc_strip_tiles (const ObmcJob & j, int seg_rows, int *strips, int *segs)
{
  const int nby_touch = std::min (j.nby, (j.h + j.yoff + j.ybsep - 1) / j.ybsep);
  *strips = (j.nbx + 1 + 15 - 1) / 15;  // (block columns 0 .. nbx: see store_lane)
  *segs = (nby_touch + seg_rows - 1) / seg_rows;
}

int
launch_obmc_strip (hipStream_t stream, const ObmcJob * d_jobs, int njobs, int total_items, int seg_rows, bool nores, uint32_t * overflow,
    int cus)
{
#ifdef SCHRO_HIP_EXPERIMENTS
  // as many waves as the device holds at once (or fewer: every wave the same number of items)
  const int slots = std::max (1, cus) * 4 * SCHRO_STRIP_WAVES;
  const int rounds = (total_items + slots - 1) / slots;
  const int waves = (total_items + rounds - 1) / rounds;
  const int wgs = (waves + kSThreads / 64 - 1) / (kSThreads / 64), total_waves = wgs * (kSThreads / 64);
  if (nores)
    SCHRO_LAUNCH ((obmc_strip_kernel < true >), dim3 (wgs), dim3 (kSThreads), 0, stream, d_jobs, njobs, total_items, total_waves, seg_rows, overflow);
  else
    SCHRO_LAUNCH ((obmc_strip_kernel < false >), dim3 (wgs), dim3 (kSThreads), 0, stream, d_jobs, njobs, total_items, total_waves, seg_rows, overflow);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "obmc (strip) launch: %s", hipGetErrorString (e));
  return 0;
#else
  (void) stream, (void) d_jobs, (void) njobs, (void) total_items, (void) seg_rows, (void) nores, (void) overflow, (void) cus;
  return set_error (SCHRO_HIP_EUNSUPPORTED, "the strip form of OBMC is built into the experiments library only");
#endif
}

}                               // namespace schro
