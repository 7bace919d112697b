// obmc_row_body.h -- OBMC + residual add + u8 clamp with the default picture weights: the row formulation, shared
// by obmc_row.hip (half- and quarter-pel references: what bench.py's headline runs), obmc_row_plain.hip (full pel:
// plain planar references, no upsample stage at all -- the reference encoder's default, schroencoder.c:4488) and
// obmc_row_eighth.hip (eighth pel).
//
// Same arithmetic and the same outer structure as obmc.hip's item kernel (a 256-thread
// workgroup owns a 128x32 output tile and a 16-bit accumulator tile in LDS; blocks are decoded
// once per tile, sorted by class, expanded into (block, row) items; blocks whose windows leave the
// image horizontally take the exact per-sample path; finish = round, add the residual, clamp,
// store) -- what changes is the
// hot loop, which r01's counters showed to be instruction issue (156 M VALU wave-instructions
// per 8 x 2160p, ~100 lane-operations per output sample):
//
//   * a lane owns a whole block ROW (12 luma pixels), not a 4-pixel segment: one item decode,
//     one address computation, one weight-table read per row instead of three;
//   * r03: the half-pel image is four planes (schro_hip_internal.h), so the samples of a block
//     row are contiguous bytes of one plane and the other taps of a quarter-pel position contiguous
//     bytes of the neighbouring planes, +-128 / +-256 bytes away in the same band of lines: the lane
//     fetches each with ONE byte-aligned buffer load of the row's length (chunks of 32 bytes that
//     advance by 16 columns: a run never leaves its chunk) -- no alignment, phase-select or
//     even / odd split instructions at all (r02: 132 of a two-reference pass's 265), and no
//     horizontal clamp either (aprons);
//   * prediction is byte-parallel: at half / quarter pel orc_combine4_nxm_u8
//     (schroorc.orc:1635-1662) degenerates to copy / 2-sample / 4-sample rounding averages,
//     v_lerp_u8 on four pixels per instruction (exact, see predict_row); a tap the window's phase
//     does not use is the first tap again (same address: lerp (a, a) = a); both references of a
//     block are blended with one more v_lerp_u8 (avgub);
//   * weights are u16 pairs (v_pk_mul_lo_u16) and two pixels share an accumulator word, so a
//     row is accumulated with xblen / 2 ds_add_u32 instead of xblen.
//
// r06 -- the forms (template parameter RK, the reference kind):
//   RK 1  half- and quarter-pel references (the tiled half-pel images), as above;
//   RK 0  mv_precision 0: the references are PLAIN planes (the decoder's own pictures: no upsample stage runs,
//         schrodecoder.c:1596-1601); a window row is one dword-aligned run + v_alignbyte, one tap; plain planes have no
//         aprons, so a window that leaves the picture (get_block lets it, up to 32 pixels: schromotion8.c:329-330)
//         joins the EDGE class, whose rows clamp every sample row and every dword of a row (predict_row_plain_abs);
//   RK 3  eighth pel: the same four taps of the tiled images with the general weights (4 - ry) (4 - rx) ..
//         ry rx of orc_combine4_nxm_u8 on 16-bit pairs (the sums stay below 4096: exact; schroframe.c:2288-2413's
//         copy and avg2 cases are the same formula).
// and NS, the segments of a block row: blocks wider than 16 samples (the 24 / 16 block set, schroparams.c:192-198)
// are decoded as NS = 2 half blocks of 12 -- a run of 24 bytes does not fit a chunk of the tiled planes, and a
// lane's row stays 3 dwords whatever the block.  Wider blocks, other picture weights and global motion stay with
// obmc.hip.
//
// References: schromotion8.c:303-335 (get_block), :542-657 (block arithmetic), :673-693
// (accumulate_slow), :700-929 (schro_motion_render_u8); schroframe.c:2111-2122 (prec 0), :2288-2482
// (schro_upsampled_frame_get_block_fast_precN); schroorc.orc:636-661 (orc_rrshift6_add_s16_2d).

#pragma once

#include "schro_hip_internal.h"
#include "obmc_common.h"
#include <algorithm>
#include <cstdlib>

namespace schro {
namespace {

constexpr int kRThreads = 256;
constexpr int kRTH = 32;                // output tile height: obmc_tiles (variants 3, 4); the kernels take it as TH (48 / 64: measured slower, HISTORY 8)
// What depends on the row length (ND dwords of prediction per block row) and on the form of the job:
//   UV = false: one plane (or the U and the V plane one after the other, NP == 2); a prediction byte is a pixel,
//     an accumulator word holds two pixels; the tile is 128 pixels wide;
//   UV = true (r04): the U and V planes of a picture from PAIR images (schro_hip_internal.h): a prediction byte
//     pair is the (U, V) sample of one pixel, an accumulator word holds that pixel's two sums (U low, V high);
//     the tile is 64 pixels wide -- the same 128 byte columns, the same pass body.
template < int ND, bool UV, int NS = 1 > struct RowGeo {
  static constexpr int kTW = UV ? 64 : 128;
  // accumulator pixels in front of the tile (+ 1 when block origins are odd): a block starts at most
  // xblen - 1 pixels in front of it.  Rows of 8 pixels and shorter (ND <= 2: 8/4 block sets, chroma planes on
  // their own; UV: at most 8 pixels = 16 bytes): 8, else 16.
  // r05, UV rows of up to 6 pixels (ND <= 3): 5 -- with 73 + 2 words a tile's tables fit an EIGHTH of a CU's LDS
  // (r06, NS > 1: a SEGMENT starts at most its own length - 1 in front of the tile; segments that do not meet the
  // tile are dropped at decode)
  static constexpr int kMargin = UV ? (ND <= 3 && NS == 1 ? 5 : 8) : (ND <= 2 ? 8 : 16);
  // accumulator row in 32-bit words: (17 + 128 + 16) / 2 -> 81, (9 + 128 + 8) / 2 -> 73, UV 8 + 64 + 8 = 80 (r05, rows of up to 6 pixels: 5 + 64 + 5 = 74) --
  // made ODD: the lanes of a pass are rows of blocks whose origins are multiples of 4 words apart, so with
  // the r02 pitch of 84 every address of an accumulate had the same word index mod 4 and 64 lanes met in 8
  // of the 32 banks (7.2 extra cycles per ds_add, simulated; 2.3 with 85)
  static constexpr int kAccW = UV ? (ND <= 3 && NS == 1 ? 75 : 81) : (ND <= 2 ? 73 : 85);
  // blocks whose footprint meets a tile and their (block, row) items (8-pixel rows and shorter are the
  // small, many blocks of chroma planes and of the 8/4 block set; a 64-pixel UV tile of 6 x 6 blocks
  // every 4 pixels meets 18 x 10 of them)
  // (for a tile of kRTH rows: the caps grow with the tile's height)
  // (UV, 6 x 6 blocks every 4 pixels: 18 x 10 blocks meet a 64 x 32 tile, 18 x 52 of their rows)
  // (NS 2: the 24 / 16 set's 10 x 4 blocks = 80 half blocks meet a luma tile, 20 x 64 of their rows at most; the sets of
  // full overlap the reference's encoder makes by default (schroengine.c:411-453: block length = 2 x separation) meet a
  // tile with more: 24 / 12 -- 13 x 5 blocks = 130 halves, its 12 / 6 chroma as (U, V) 13 x 8 = 208 halves, 26 columns x 2
  // blocks over each of 32 rows = 1664 items; 32 / 16 -- 80 halves of 16 pixels, 1280 items.  ND 4, 16 / 8: 18 columns x 64)
  static constexpr int kBlk = ND <= 2 ? 344 : (NS > 1 ? (UV ? 208 : 144) : UV ? (ND == 3 ? 180 : 192) : 128) * kRTH / 32,
      kItem = ND <= 2 ? 1792 : (NS > 1 ? 1664 : UV ? (ND == 3 ? 960 : 1152) : (ND == 3 ? 1024 : 1152)) * kRTH / 32;
  // (row, segment, pixel pair | UV: pixel) weight words: 2 * ND per row and segment (zero beyond the block), 32 rows
  static constexpr int kWRow = 2 * ND * NS;
  static constexpr int kWCap = 32 * kWRow;
  // folded x weights: words per edge type (4 types); the ramps' lengths
  static constexpr int kXF = NS == 1 ? 8 : kWRow, kWxN = NS == 1 ? 16 : 32;
  static constexpr bool kPadBlk = UV || ND > 2;        // block records of nine words (see RowBlkT)
  // r05: the weight table of a plane geometry, made on the HOST (obmc_row.hip: obmc_row_weight_table) and copied into
  // LDS by the tile: kWCap (row, pair) words, 4 x kXF folded x pairs, 128 folded y weights, the two ramps (kWxN + 32 ints)
  static constexpr int kWFoldY = kWCap + 4 * kXF, kWRampX = kWFoldY + 128, kWRampY = kWRampX + kWxN;
  static constexpr int kWTab = kWRampY + 32;
};
// Item classes (one straight-line pass body each): both references / the first / the second / DC /
// edge (windows clamped vertically and / or folded weights, any mode: still a row per lane) / rim (DC
// values outside 8 bits, geometries beyond the weight table: per sample).  Inside the reference
// classes the items are SORTED by which taps their windows need -- slot = class base + (X + 1 tap |
// Y + 1 taps << 1) per reference -- so the 64 items of a pass mostly agree, and a pass fetches a tap
// only if one of its lanes needs it (wave-uniform branches on ballots): at quarter pel a window
// needs 1, 2 or 4 of the 4 taps with probability 1/4, 1/2, 1/4.
constexpr int kRBoth = 0, kRRef0 = 1, kRRef1 = 2, kRDc = 3, kREdge = 4, kRRim = 5;
constexpr int kRSlots = 16 + 4 + 4 + 3;
__host__ __device__ constexpr int
row_slot_base (int cls)
{
  return cls == kRBoth ? 0 : cls == kRRef0 ? 16 : cls == kRRef1 ? 20 : cls == kRDc ? 24 : cls == kREdge ? 25 : 26;
}

typedef unsigned short u16x2 __attribute__ ((ext_vector_type (2)));
typedef short s16x2 __attribute__ ((ext_vector_type (2)));

// one reference's window of a block
struct __attribute__ ((aligned (4))) RowRef {
  int base;                     // byte offset of the window's first sample inside a band of lines: chunk, byte in the
                                // chunk, plane (edge class: without the row-parity part of the plane; rim: fx | fy << 16)
  uint32_t ydb;                 // plane row of that sample (edge class: its half-pel row, signed) | dB << 16: the X + 1 taps
                                // are dB bytes on (+128: the other column parity; 1 - 128: and one column on; 0: not used)
  int dci;                      // dC << 16 | inc: the Y + 1 taps are dC bytes and inc plane rows on (+256, 0 | -256, 1 | 0, 0)
};

// 36 bytes: a pitch of NINE words.  The lanes of a pass read the records of the 5 - 11 blocks their items
// belong to; with r03's 32-byte records (eight words) blocks four apart shared their banks
// (SQ_LDS_BANK_CONFLICT 11.9 M -> 20.1 M cycles per 8 x 2160p step against the 40-byte records before);
// an odd pitch maps 32 consecutive blocks to 32 different banks.
// (The 6-pixel-row kernels of chroma planes on their own keep 32 bytes: 344 records, and the 1.4 KB more
// would cost them the sixth workgroup per CU.)
template < bool PAD > struct alignas (PAD ? 4 : 8) RowBlkT {
  int16_t y, x;                 // block origin relative to the tile
  uint32_t fr;                  // first block row inside the tile | rows inside << 8 | flags << 16:
                                // flag bits 0-1 mode, 2-5 weights fold at top | bottom | left | right,
                                // 6 + r: reference r's window at a vertical quarter position (edge class)
  RowRef r[2];                  // mode 0 (no reference used): r[0].base = the DC values of the job's planes,
                                // 16 bits each (first plane low)
  uint32_t pad[PAD ? 1 : 0];
};
static_assert (sizeof (RowBlkT < true >) == 36 && sizeof (RowBlkT < false >) == 32, "block records: an odd number of words / r03's");

template < typename B >
__device__ __forceinline__ uint32_t
blk_flags (const B & hb)
{
  return hb.fr >> 16;
}

template < typename B >
__device__ __forceinline__ uint32_t
blk_ry (const B & hb, int r)
{
  return (hb.fr >> (16 + 6 + r)) & 1u;
}

template < typename B >
__device__ __forceinline__ int
blk_dc (const B & hb, int pl)
{
  return pl ? hb.r[0].base >> 16 : (int) (int16_t) hb.r[0].base;
}


// what differs between the planes of a job; kept in LDS (one copy per workgroup) so that the
// pointers of the plane not being worked on cost no scalar registers -- with both planes'
// pointers live beside the block geometry the kernel spilled 119 SGPRs
struct __attribute__ ((aligned (8))) PlaneIO {
  const uint8_t *ref[2];
  const void *residual;
  uint8_t *out;
  int residual_stride, out_stride;
};

__device__ __forceinline__ uint32_t
lerp1 (uint32_t a, uint32_t b)
{
  return __builtin_amdgcn_lerp (a, b, 0x01010101u);     // per byte (a + b + 1) >> 1 = avgub
}

__device__ __forceinline__ void
acc_add_exact (uint32_t * word, int high, uint32_t value)
{
  // 16-bit wrapping add inside one half of the word (a DC value outside 0..255: the
  // reference's s16 sum wraps and must not carry into the neighbour pixel)
  unsigned int old = __hip_atomic_load (word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), assumed;
  do {
    assumed = old;
    const unsigned int upd = high ? (assumed & 0xffffu) | ((assumed + (value << 16)) & 0xffff0000u)
        : (assumed & 0xffff0000u) | ((assumed + value) & 0xffffu);
    old = atomicCAS (word, assumed, upd);
  } while (old != assumed);
}

// the accumulator word and half of tile-relative pixel (x, y); `par` = 1 when block origins are odd.
// UV: the pixel's word; its halves are the two components
template < typename G, bool UV >
__device__ __forceinline__ uint32_t *
acc_word (uint32_t * acc, int par, int x, int y, int *half)
{
  if constexpr (UV) {
    *half = 0;
    return acc + y * G::kAccW + (x + G::kMargin);
  } else {
    const int idx = x + G::kMargin + par;
    *half = idx & 1;
    return acc + y * G::kAccW + (idx >> 1);
  }
}

// ---- one reference's prediction of a block row: ND dwords of 4 pixels ---------------------
// ND dwords from byte offset `off` of the reference (any alignment; beyond the buffer: zeros).
// The load itself is dword-aligned and one dword longer, the bytes are shifted into place with
// v_alignbyte: a byte-aligned load of n dwords takes the texture path n times as long as a
// dword-aligned one (scripts/ta_rate_bench.hip: 16 / 48 cycles per wave for 12 bytes per lane), and
// with one such load per tap the passes were bound by exactly that (TA busy 76 %).
template < int ND > struct RawRun {
  uint32_t c[ND + 1];           // the dwords from the dword-aligned address on
  uint32_t sh;                  // where the run starts in the first one
};

// scratch builds (experiments only): the gather's cache policy -- -DSCHRO_ROW_GATHER_AUX=2 non-temporal, 16 agent scope (sc1:
// every load misses the CU's own cache), 17 sc0 | sc1
#if defined (SCHRO_HIP_EXPERIMENTS) && defined (SCHRO_ROW_GATHER_AUX)
constexpr int kGatherAux = SCHRO_ROW_GATHER_AUX;
#else
constexpr int kGatherAux = 0;
#endif

template < int ND >
__device__ __forceinline__ void
issue_run (__amdgpu_buffer_rsrc_t ref, uint32_t off, RawRun < ND > &r)
{
  const uint32_t al = off & ~3u;
  r.sh = off & 3u;
  if constexpr (ND == 1) {
    const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64 (ref, (int) al, 0, kGatherAux);
    r.c[0] = q.x;
    r.c[1] = q.y;
  } else if constexpr (ND == 2) {
    typedef uint32_t u32x3 __attribute__ ((ext_vector_type (3)));
    const u32x3 q = __builtin_amdgcn_raw_buffer_load_b96 (ref, (int) al, 0, kGatherAux);
    r.c[0] = q.x;
    r.c[1] = q.y;
    r.c[2] = q.z;
  } else if constexpr (ND == 3) {
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128 (ref, (int) al, 0, kGatherAux);
    r.c[0] = q.x;
    r.c[1] = q.y;
    r.c[2] = q.z;
    r.c[3] = q.w;
  } else {
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128 (ref, (int) al, 0, kGatherAux);
    const u32x2 q2 = __builtin_amdgcn_raw_buffer_load_b64 (ref, (int) al + 16, 0, kGatherAux);       // (8 bytes: as fast as 4)
    r.c[0] = q.x;
    r.c[1] = q.y;
    r.c[2] = q.z;
    r.c[3] = q.w;
    r.c[4] = q2.x;
  }
}

template < int ND >
__device__ __forceinline__ void
align_run (const RawRun < ND > &r, uint32_t * d)
{
#pragma unroll
  for (int k = 0; k < ND; k++)
    d[k] = __builtin_amdgcn_alignbyte (r.c[k + 1], r.c[k], r.sh);
}

// byte offset of plane row y inside the image: band (4 rows) * stride + 32 * row in the band
__device__ __forceinline__ uint32_t
row_ofs (uint32_t y, uint32_t stride)
{
  return __umul24 (y >> 2, stride) + ((y & 3u) << 5);
}

// A tap the window's phase does not use has the first tap's address (dB / dC 0): the same line
// again, and lerp (a, a) = a -- and is not fetched at all when no lane of the pass uses it (the items
// are sorted by that, see the slots).
// ABS (edge class): rr.ydb holds the window's first HALF-PEL row, signed, and every sample row is
// clamped to the image on its own (fetch_ref's CLAMP on y): the row's parity picks the plane
// W (r06, eighth pel): bits 4-5 / 6-7 of rr.dci are the window's position rx / ry inside its half-pel cell (0 .. 3) and
// the four taps are blended with orc_combine4_nxm_u8's weights (4 - ry) (4 - rx), (4 - ry) rx, ry (4 - rx), ry rx,
// + 8, >> 4 (schroorc.orc:1635-1662; schroframe.c:2288-2413's copy and avg2 cases are the same formula) on pairs of
// 16-bit sums: a sum stays below 16 * 255 + 8, so two of them share a dword and a 24-bit multiply
template < int ND, bool ABS = false, bool W = false >
__device__ __forceinline__ void
predict_row (const ObmcJob & job, __amdgpu_buffer_rsrc_t ref, uint32_t stride, const RowRef & rr, uint32_t ry, int row, uint32_t * out)
{
  const int dB = (int) rr.ydb >> 16;
  uint32_t offA, offC = 0;
  bool any_b, any_c;
  if constexpr (ABS) {
    const int hy = (int) (int16_t) (rr.ydb & 0xffffu) + 2 * row, gh = 2 * job.h - 2;
    const uint32_t Y0 = (uint32_t) clampi (hy, 0, gh), Y1 = (uint32_t) clampi (hy + (int) ry, 0, gh);
    offA = (uint32_t) rr.base + ((Y0 & 1u) << 8) + row_ofs (Y0 >> 1, stride);
    offC = (uint32_t) rr.base + ((Y1 & 1u) << 8) + row_ofs (Y1 >> 1, stride);
    any_b = any_c = true;
  } else {
    const uint32_t y = (rr.ydb & 0xffffu) + (uint32_t) row;
    offA = (uint32_t) rr.base + row_ofs (y, stride);
    any_b = __ballot (dB != 0) != 0;
    any_c = __ballot ((W ? rr.dci & ~0xf0 : rr.dci) != 0) != 0;
    if (any_c)
      offC = (uint32_t) (rr.base + (rr.dci >> 16)) + row_ofs (y + ((uint32_t) rr.dci & 1u), stride);
  }
  // every load of the pass goes out before the first result is touched (a use in front of a branch
  // makes the compiler wait there: two round trips per reference instead of one)
  RawRun < ND > qa, qb, qc, qd;
  issue_run < ND > (ref, offA, qa);
  if (any_b)
    issue_run < ND > (ref, offA + (uint32_t) dB, qb);
  if (any_c) {
    issue_run < ND > (ref, offC, qc);
    if (any_b)
      issue_run < ND > (ref, offC + (uint32_t) dB, qd);
  }
  uint32_t a[ND];
  align_run < ND > (qa, a);
  if constexpr (W) {
    const uint32_t rx = ((uint32_t) rr.dci >> 4) & 3u, rw = ((uint32_t) rr.dci >> 6) & 3u;
    const uint32_t w00 = (4u - rw) * (4u - rx), w01 = (4u - rw) * rx, w10 = rw * (4u - rx), w11 = rw * rx;
    uint32_t lo[ND], hi[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) {
      lo[k] = __umul24 (__builtin_amdgcn_perm (0u, a[k], 0x0c010c00u), w00) + 0x00080008u;
      hi[k] = __umul24 (__builtin_amdgcn_perm (0u, a[k], 0x0c030c02u), w00) + 0x00080008u;
    }
    auto add_tap = [&] (const RawRun < ND > &q, uint32_t w) {
      uint32_t t[ND];
      align_run < ND > (q, t);
#pragma unroll
      for (int k = 0; k < ND; k++) {
        lo[k] += __umul24 (__builtin_amdgcn_perm (0u, t[k], 0x0c010c00u), w);
        hi[k] += __umul24 (__builtin_amdgcn_perm (0u, t[k], 0x0c030c02u), w);
      }
    };
    if (any_b)
      add_tap (qb, w01);
    if (any_c) {
      add_tap (qc, w10);
      if (any_b)
        add_tap (qd, w11);
    }
    // (a shift of the dword: what a sum's low four bits leave in its neighbour's top bits is not picked up)
#pragma unroll
    for (int k = 0; k < ND; k++)
      out[k] = __builtin_amdgcn_perm (hi[k] >> 4, lo[k] >> 4, 0x06040200u);
    return;
  }
  if (!any_c) {
    if (!any_b) {
#pragma unroll
      for (int k = 0; k < ND; k++)
        out[k] = a[k];
    } else {
      uint32_t b[ND];
      align_run < ND > (qb, b);
#pragma unroll
      for (int k = 0; k < ND; k++)
        out[k] = lerp1 (a[k], b[k]);
    }
  } else if (!any_b) {
    uint32_t c[ND];
    align_run < ND > (qc, c);
#pragma unroll
    for (int k = 0; k < ND; k++)
      out[k] = lerp1 (a[k], c[k]);      // (a + a + c + c + 2) >> 2
  } else {
    uint32_t b[ND], c[ND], d[ND];
    align_run < ND > (qb, b);
    align_run < ND > (qc, c);
    align_run < ND > (qd, d);
#pragma unroll
    for (int k = 0; k < ND; k++) {
      // per byte (a + b + c + d + 2) >> 2 exactly: with c1 = (a+b+1)>>1, c2 = (c+d+1)>>1 and l = the
      // bit an average rounded up by, (c1 + c2 + 1 - (l1 | l2)) >> 1; for a + a + c + c: (a + c + 1) >> 1
      const uint32_t h0 = lerp1 (a[k], b[k]), h1 = lerp1 (c[k], d[k]);
      out[k] = __builtin_amdgcn_lerp (h0, h1, ~((a[k] ^ b[k]) | (c[k] ^ d[k])));
    }
  }
}

// scratch builds (experiments only, -DSCHRO_ROW_BOTH_AT_ONCE): predict_row's two halves apart -- every load of BOTH references of a
// two-reference item goes out before either prediction is made (half the round trips of such a pass, twice the registers in flight)
template < int ND > struct TapRuns {
  RawRun < ND > qa, qb, qc, qd;
  bool any_b, any_c;
};

template < int ND >
__device__ __forceinline__ void
issue_taps (__amdgpu_buffer_rsrc_t ref, uint32_t stride, const RowRef & rr, int row, TapRuns < ND > &t)
{
  const int dB = (int) rr.ydb >> 16;
  const uint32_t y = (rr.ydb & 0xffffu) + (uint32_t) row;
  const uint32_t offA = (uint32_t) rr.base + row_ofs (y, stride);
  t.any_b = __ballot (dB != 0) != 0;
  t.any_c = __ballot (rr.dci != 0) != 0;
  issue_run < ND > (ref, offA, t.qa);
  if (t.any_b)
    issue_run < ND > (ref, offA + (uint32_t) dB, t.qb);
  if (t.any_c) {
    const uint32_t offC = (uint32_t) (rr.base + (rr.dci >> 16)) + row_ofs (y + ((uint32_t) rr.dci & 1u), stride);
    issue_run < ND > (ref, offC, t.qc);
    if (t.any_b)
      issue_run < ND > (ref, offC + (uint32_t) dB, t.qd);
  }
}

template < int ND >
__device__ __forceinline__ void
finish_taps (const TapRuns < ND > &t, uint32_t * out)
{
  uint32_t a[ND];
  align_run < ND > (t.qa, a);
  if (!t.any_c) {
    if (!t.any_b) {
#pragma unroll
      for (int k = 0; k < ND; k++)
        out[k] = a[k];
    } else {
      uint32_t b[ND];
      align_run < ND > (t.qb, b);
#pragma unroll
      for (int k = 0; k < ND; k++)
        out[k] = lerp1 (a[k], b[k]);
    }
  } else if (!t.any_b) {
    uint32_t c[ND];
    align_run < ND > (t.qc, c);
#pragma unroll
    for (int k = 0; k < ND; k++)
      out[k] = lerp1 (a[k], c[k]);
  } else {
    uint32_t b[ND], c[ND], d[ND];
    align_run < ND > (t.qb, b);
    align_run < ND > (t.qc, c);
    align_run < ND > (t.qd, d);
#pragma unroll
    for (int k = 0; k < ND; k++) {
      const uint32_t h0 = lerp1 (a[k], b[k]), h1 = lerp1 (c[k], d[k]);
      out[k] = __builtin_amdgcn_lerp (h0, h1, ~((a[k] ^ b[k]) | (c[k] ^ d[k])));
    }
  }
}

// r06, RK 0 -- a block row from a PLAIN plane (mv_precision 0: schroframe.c:2111-2122, one tap): rr.base = the window's
// first column, rr.ydb = its first row; a window of this class lies inside the picture
template < int ND >
__device__ __forceinline__ void
predict_row_plain (__amdgpu_buffer_rsrc_t ref, uint32_t stride, const RowRef & rr, int row, uint32_t * out)
{
  RawRun < ND > qa;
  issue_run < ND > (ref, (uint32_t) rr.base + __umul24 ((rr.ydb & 0xffffu) + (uint32_t) row, stride), qa);
  align_run < ND > (qa, out);
}

// ... and of the edge class: the window may leave the picture by up to 32 pixels on any side (get_block's clamp,
// schromotion8.c:329-330; the reference reads its frames' replicated aprons there, schroframe.c:1940-1998) -- the
// sample row is clamped, and every dword of the row comes from the four picture bytes nearest to it, put in
// place by a byte permute: out byte b of dword k = plane [clamp (fx + 4 k + b, 0, w - 1)].
// rr.base: the window's first column, signed; rr.ydb: its first row, signed 16 bits
template < int ND >
__device__ __forceinline__ void
predict_row_plain_abs (const ObmcJob & job, __amdgpu_buffer_rsrc_t ref, uint32_t stride, const RowRef & rr, int row, uint32_t * out)
{
  const int Y = clampi ((int) (int16_t) (rr.ydb & 0xffffu) + row, 0, job.h - 1);
  const uint32_t rofs = __umul24 ((uint32_t) Y, stride);
  u32x2 q[ND];
  uint32_t sh[ND];
  int d[ND];
#pragma unroll
  for (int k = 0; k < ND; k++) {
    const int t = rr.base + 4 * k, xs = clampi (t, 0, job.w - 4);
    const uint32_t off = rofs + (uint32_t) xs;
    d[k] = t - xs;              // < 0: the dword hangs over the left edge, > 0: over the right one
    sh[k] = off & 3u;
    q[k] = __builtin_amdgcn_raw_buffer_load_b64 (ref, (int) (off & ~3u), 0, 0);
  }
#pragma unroll
  for (int k = 0; k < ND; k++) {
    const uint32_t v = __builtin_amdgcn_alignbyte (q[k].y, q[k].x, sh[k]);
    const uint32_t sel = (uint32_t) clampi (d[k], 0, 3) | ((uint32_t) clampi (d[k] + 1, 0, 3) << 8)
        | ((uint32_t) clampi (d[k] + 2, 0, 3) << 16) | ((uint32_t) clampi (d[k] + 3, 0, 3) << 24);
    out[k] = __builtin_amdgcn_perm (0u, v, sel);
  }
}

// r06 -- (U, V) jobs at full pel: the references are two PLAIN planes per picture; a block row of n samples is n bytes
// of each, fetched one after the other and interleaved into the (U, V) byte pairs the rest of the pass works on
// (v_perm), so the accumulator, the weights and the finish are the pair images' (one decode and one tile for both planes)
template < int ND >
__device__ __forceinline__ void
interleave_uv (const uint32_t * u, const uint32_t * v, uint32_t * out)
{
#pragma unroll
  for (int k = 0; k < ND; k++)
    out[k] = __builtin_amdgcn_perm (v[k >> 1], u[k >> 1], (k & 1) ? 0x07030602u : 0x05010400u);
}

// one reference's prediction of a block row in the kernel's form; `ry`: edge class, RK 1: the window's rows lie at a
// vertical quarter position.  ref_b: the V plane of a (U, V) job on plain planes
template < int ND, int RK, bool ABS, bool UV >
__device__ __forceinline__ void
predict (const ObmcJob & job, __amdgpu_buffer_rsrc_t ref, __amdgpu_buffer_rsrc_t ref_b, uint32_t stride, const RowRef & rr, uint32_t ry, int row,
    uint32_t * out)
{
  if constexpr (RK == 0 && UV) {
    constexpr int NH = (ND + 1) / 2;    // dwords of each plane: ND dwords of (U, V) pairs are 2 ND samples
    uint32_t u[NH], v[NH];
    if constexpr (ABS) {
      predict_row_plain_abs < NH > (job, ref, stride, rr, row, u);
      predict_row_plain_abs < NH > (job, ref_b, stride, rr, row, v);
    } else {
      // (both loads out before either is touched)
      RawRun < NH > qu, qv;
      const uint32_t off = (uint32_t) rr.base + __umul24 ((rr.ydb & 0xffffu) + (uint32_t) row, stride);
      issue_run < NH > (ref, off, qu);
      issue_run < NH > (ref_b, off, qv);
      align_run < NH > (qu, u);
      align_run < NH > (qv, v);
    }
    interleave_uv < ND > (u, v, out);
  } else if constexpr (RK == 0) {
    if constexpr (ABS)
      predict_row_plain_abs < ND > (job, ref, stride, rr, row, out);
    else
      predict_row_plain < ND > (ref, stride, rr, row, out);
  } else if constexpr (RK == 3) {
    predict_row < ND, ABS, true > (job, ref, stride, rr, ABS ? (((uint32_t) rr.dci & 0xc0u) ? 1u : 0u) : 0u, row, out);
  } else {
    predict_row < ND, ABS > (job, ref, stride, rr, ry, row, out);
  }
}

// r06 -- the two references of a block blended with the PICTURE WEIGHTS, for weights that are not negative and add up to
// 1 << bits (a fade): per byte (w1 a + w2 b + ((1 << bits) >> 1)) >> bits.  For such weights the reference's two block
// arithmetics agree -- interior blocks: mullw by w << (6 - bits), addw, + 32, shrsw 6 (block_acc_biref, schromotion8.c:131-163);
// edge blocks: orc_combine2_nxm_u8 (schroorc.orc:1737-1757), whose clamp never acts -- and a block of one reference is
// its prediction unchanged (oneref_noscale, :391-397, :44-73); with 1, 1 / 2 it is avgub.  Everything else (a gain, a
// negative weight) stays with obmc.hip, which keeps the two arithmetics apart.  Two 16-bit sums share a dword (< 2^15).
__device__ __forceinline__ uint32_t
blend_weighted (uint32_t a, uint32_t b, uint32_t w1, uint32_t w2, uint32_t round2, uint32_t bits)
{
  const uint32_t lo = __umul24 (__builtin_amdgcn_perm (0u, a, 0x0c010c00u), w1) + __umul24 (__builtin_amdgcn_perm (0u, b, 0x0c010c00u), w2) + round2;
  const uint32_t hi = __umul24 (__builtin_amdgcn_perm (0u, a, 0x0c030c02u), w1) + __umul24 (__builtin_amdgcn_perm (0u, b, 0x0c030c02u), w2) + round2;
  // (a shift of the dword: what a sum's low bits leave in its neighbour's top bits is not picked up)
  return __builtin_amdgcn_perm (hi >> bits, lo >> bits, 0x06040200u);
}

// one pass: every lane predicts one (block, row) item and adds it into the accumulator tile
struct RowRefs {
  __amdgpu_buffer_rsrc_t rsrc[2];       // the plane's references as buffers: whole bands of 4 rows
  uint32_t stride[2];
  __amdgpu_buffer_rsrc_t rsrc_b[2];     // r06, (U, V) jobs on plain planes (RK 0): the V planes (rsrc: the U planes), same strides
};

// the DC value(s) of a block as prediction bytes: a plane's byte four times, UV: (U, V) twice
template < bool UV, typename B >
__device__ __forceinline__ uint32_t
dc_bytes (const B & hb, int pl)
{
  if constexpr (UV)
    return ((uint32_t) (blk_dc (hb, 0) & 0xff) | ((uint32_t) (blk_dc (hb, 1) & 0xff) << 8)) * 0x00010001u;
  else
    return (uint32_t) (blk_dc (hb, pl) & 0xff) * 0x01010101u;
}

template < int ND, bool UV, int CLS, bool EXACT, int RK, int NS, bool WP, bool PAD = false >
__device__ __forceinline__ void
row_pass (const ObmcJob & job, int pl, const RowRefs & refs, const uint16_t * s_item, const RowBlkT < RowGeo < ND, UV, NS >::kPadBlk > *s_hot,
    const uint32_t * s_wp, uint32_t * acc, int par, int it, int hi)
{
  typedef RowGeo < ND, UV, NS > G;
  const int e = s_item[min (it, hi - 1)];
  const auto & hb = s_hot[e & 0x1ff];
  // (PAD: bit 15 marks a lane that only keeps its quad on the line of its neighbours -- it loads one of their rows and adds nothing)
  const int row = PAD ? (e >> 9) & 63 : e >> 9;
  // WP: picture weights other than 1, 1 / 2 (obmc_row_form admits the non-negative ones that add up to 1 << bits) -- kernels of
  // their own, so that the default weights' kernels do not carry the weights in their scalar registers
  constexpr bool weighted = WP;
  const uint32_t wround = WP ? (uint32_t) ((1 << job.wbits) >> 1) * 0x00010001u : 0u;
  uint32_t p[ND];
  if constexpr (CLS == kRDc) {
    // (DC values outside 0..255 are not in this class: rim)
#pragma unroll
    for (int k = 0; k < ND; k++)
      p[k] = dc_bytes < UV > (hb, pl);
  } else if constexpr (CLS == kREdge) {
    // any mode: both references are read (an unused one at offset 0) and the mode selects
    uint32_t p1[ND];
    predict < ND, RK, true, UV > (job, refs.rsrc[0], refs.rsrc_b[0], refs.stride[0], hb.r[0], blk_ry (hb, 0), row, p);
    __builtin_amdgcn_sched_barrier (0);
    predict < ND, RK, true, UV > (job, refs.rsrc[1], refs.rsrc_b[1], refs.stride[1], hb.r[1], blk_ry (hb, 1), row, p1);
    const uint32_t mode = blk_flags (hb) & 3u;
    const uint32_t dc = dc_bytes < UV > (hb, pl);       // (meaningful in mode 0 only)
    const uint32_t m0 = (mode & 1u) ? 0xffffffffu : 0u, m1 = (mode & 2u) ? 0xffffffffu : 0u;
#pragma unroll
    for (int k = 0; k < ND; k++) {
      const uint32_t a = (p[k] & m0) | (p1[k] & ~m0), b = (p1[k] & m1) | (p[k] & ~m1);  // one reference: average it with itself
      p[k] = mode ? (weighted ? blend_weighted (a, b, (uint32_t) job.w1, (uint32_t) job.w2, wround, (uint32_t) job.wbits) : lerp1 (a, b)) : dc;
    }
  } else if constexpr (CLS == kRBoth) {
    uint32_t p1[ND];
#if defined (SCHRO_HIP_EXPERIMENTS) && defined (SCHRO_ROW_BOTH_AT_ONCE)
    if constexpr (RK == 1) {
      TapRuns < ND > t0, t1;
      issue_taps < ND > (refs.rsrc[0], refs.stride[0], hb.r[0], row, t0);
      issue_taps < ND > (refs.rsrc[1], refs.stride[1], hb.r[1], row, t1);
      __builtin_amdgcn_sched_barrier (0);       // (or the scheduler moves the second reference's loads behind the first one's use)
      finish_taps < ND > (t0, p);
      finish_taps < ND > (t1, p1);
    } else
#endif
    {
    predict < ND, RK, false, UV > (job, refs.rsrc[0], refs.rsrc_b[0], refs.stride[0], hb.r[0], 0u, row, p);
    __builtin_amdgcn_sched_barrier (0); // one reference at a time: half the registers in flight (both at once: measured slower)
    predict < ND, RK, false, UV > (job, refs.rsrc[1], refs.rsrc_b[1], refs.stride[1], hb.r[1], 0u, row, p1);
    }
    if (weighted) {             // (a uniform branch: the job's weights)
#pragma unroll
      for (int k = 0; k < ND; k++)
        p[k] = blend_weighted (p[k], p1[k], (uint32_t) job.w1, (uint32_t) job.w2, wround, (uint32_t) job.wbits);
    } else {
#pragma unroll
      for (int k = 0; k < ND; k++)
        p[k] = lerp1 (p[k], p1[k]);     // avgub of the two predictions, schromotion8.c:560-566 with the default weights
    }
  } else {
    constexpr int r = CLS == kRRef1 ? 1 : 0;
    predict < ND, RK, false, UV > (job, refs.rsrc[r], refs.rsrc_b[r], refs.stride[r], hb.r[r], 0u, row, p);
  }
  if (it >= hi || (PAD && (e & 0x8000)))
    return;
  // scratch builds (experiments only): the passes with their loads alone -- the prediction goes into ONE accumulator word
  // and nothing else is done with it --, VERDICT r05 "what's weak" 2: is the gather by itself the launch?
#if defined (SCHRO_HIP_EXPERIMENTS) && defined (SCHRO_ROW_PASS_LOADS_ONLY)
  {
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < ND; k++)
      x ^= p[k];
    atomicAdd (acc + (it & 1023), x);
    return;
  }
#endif
  int half;
  uint32_t *aw = acc_word < G, UV > (acc, par, hb.x, hb.y + row, &half);       // (block origins + par are even: half == 0)
  // the row's 2 * ND weight pairs (zero beyond the block: no tests in the loop), read in one go.  A word's two
  // 16-bit weights multiply the two bytes of a prediction byte pair: two neighbouring pixels, UV: the pixel's U and V
  // (NS > 1: the words of the block's segment this record stands for)
  const int seg_w = NS > 1 ? (int) ((blk_flags (hb) >> 8) & 1u) * 2 * ND : 0;
  uint32_t w[2 * ND];
  if constexpr (CLS == kREdge) {
    // weights folded at the picture's rim (schromotion8.c:673-693): 1-D tables per edge type behind
    // the plain products -- (left | right << 1) pairs of x weights, (top | bottom << 1) y weights
    const uint32_t fb = (blk_flags (hb) >> 2) & 15u;
    const uint32_t *wxf = s_wp + G::kWCap + G::kXF * (fb >> 2) + seg_w, *wyf = s_wp + G::kWFoldY + 32 * (fb & 3u);
    const uint32_t wy2 = wyf[row] * 0x00010001u;
#pragma unroll
    for (int k = 0; k < 2 * ND; k++)
      w[k] = __builtin_bit_cast (uint32_t, (u16x2) (__builtin_bit_cast (u16x2, wxf[k]) * __builtin_bit_cast (u16x2, wy2)));
  } else {
    const u32x2 *wp = reinterpret_cast < const u32x2 * >(s_wp + G::kWRow * row + seg_w);
#pragma unroll
    for (int k = 0; k < ND; k++) {
      const u32x2 q = wp[k];
      w[2 * k] = q.x;
      w[2 * k + 1] = q.y;
    }
  }
#pragma unroll
  for (int k = 0; k < 2 * ND; k++) {
    const uint32_t px = __builtin_amdgcn_perm (0u, p[k >> 1], (k & 1) ? 0x0c030c02u : 0x0c010c00u);
    const uint32_t v = __builtin_bit_cast (uint32_t, (u16x2) (__builtin_bit_cast (u16x2, px) * __builtin_bit_cast (u16x2, w[k])));
    if constexpr (EXACT) {
      acc_add_exact (aw + k, 0, v & 0xffffu);
      acc_add_exact (aw + k, 1, v >> 16);
    } else {
      atomicAdd (aw + k, v);    // sums of pred * weight <= 255 * 64: no carry between the halves
    }
  }
}

// the passes of one class; *turn counts the passes of the classes before it, so that the four
// waves take the tile's passes in turn whatever the class sizes (with nine classes most have one
// or two passes: "wave w takes the w-th pass of every class" left wave 0 with nine passes and
// wave 3 with none)
template < int ND, bool UV, int CLS, int RK, int NS, bool WP, bool PAD = false >
__device__ __forceinline__ void
row_class (const ObmcJob & job, int pl, const RowRefs & refs, const uint16_t * s_item, const RowBlkT < RowGeo < ND, UV, NS >::kPadBlk > *s_hot,
    const uint32_t * s_wp, uint32_t * acc, int par, int lo, int hi, bool exact, int *turn)
{
  // (wave-uniform values in scalar registers: the class loops are scalar branches, not exec masks)
  const int wave = __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int npass = (hi - lo + 63) >> 6;
  constexpr int kWaves = kRThreads / 64;
  const int k0 = (wave - *turn) & (kWaves - 1);
  *turn = (*turn + npass) & (kWaves - 1);
  // scratch builds (experiments only): the waves inside their passes at a higher issue priority than the waves that set a
  // tile up (-DSCHRO_ROW_PRIO_PASS=3), or the other way round (-DSCHRO_ROW_PRIO_PASS=0 -DSCHRO_ROW_PRIO_SETUP=3)
#if defined (SCHRO_HIP_EXPERIMENTS) && defined (SCHRO_ROW_PRIO_PASS)
  __builtin_amdgcn_s_setprio (SCHRO_ROW_PRIO_PASS);
#endif
  if (exact) {                  // a DC value outside 0..255 somewhere in the tile: rare, kept out of the hot loop
    for (int k = k0; k < npass; k += kWaves)
      row_pass < ND, UV, CLS, true, RK, NS, WP, PAD > (job, pl, refs, s_item, s_hot, s_wp, acc, par, lo + 64 * k + lane, hi);
  } else {
    for (int k = k0; k < npass; k += kWaves)
      row_pass < ND, UV, CLS, false, RK, NS, WP, PAD > (job, pl, refs, s_item, s_hot, s_wp, acc, par, lo + 64 * k + lane, hi);
  }
}

// picture-rim block rows: per-sample clamp and weight folding (accumulate_slow), 4 pixels.
// UV: of component cb of the pair images, into that half of the pixels' words
template < int PC, typename G, bool UV >
__device__ __forceinline__ void
row_slow (const ObmcJob & job, const PlaneIO & io, int cb, int bx, int by, int md, const int *fx, const int *fy, int row, int seg,
    int x_lo, int y_lo, int xfold_hi, int yfold_hi, const int *s_wx, const int *s_wy, uint32_t * acc, int par, bool exact)
{
  constexpr int ps = UV ? 1 : 0;
  const uint8_t *const refs[2] = { io.ref[0], io.ref[1] };
  const int prec = job.prec;
  const int y = by + row, xs = bx + 4 * seg;
  const int mode = md & 3;
  int pred[4];
  if (mode == 0) {
    pred[0] = pred[1] = pred[2] = pred[3] = md >> 8;
  } else {
    int val[2][4] = { {0, 0, 0, 0}, {0, 0, 0, 0} };
#pragma unroll
    for (int r = 0; r < 2; r++) {
      if (!(mode & (r + 1)))
        continue;
      if constexpr (PC == 2) {
        // fetch_ref < 2 > for four pixels at once: they are 8 eighth-pels apart, share the blend
        // weights and read nine adjacent half-pel columns of two rows, each clamped on its own;
        // samples whose weight is zero (integer positions) are not fetched
        const int sx = fx[r] + 4 * seg * (1 << prec), sy = fy[r] + row * (1 << prec);
        const int x8 = prec == 2 ? sx * 2 : sx, y8 = prec == 2 ? sy * 2 : sy;
        const int hx = x8 >> 2, hy = y8 >> 2, rx = x8 & 3, ry = y8 & 3;
        const int Y0 = clampi (hy, 0, 2 * job.h - 2), Y1 = clampi (hy + 1, 0, 2 * job.h - 2);
        int p0[9], p1[9];
#pragma unroll
        for (int j = 0; j < 9; j++) {
          const int X = clampi (hx + j, 0, 2 * job.w - 2);
          const bool need = (j & 1) == 0 || rx != 0;
          p0[j] = need ? (int) gload < uint8_t > (refs[r] + hp_offset (X, Y0, job.ref_stride[r], ps, cb)) : 0;
          p1[j] = need && ry != 0 ? (int) gload < uint8_t > (refs[r] + hp_offset (X, Y1, job.ref_stride[r], ps, cb)) : 0;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int v = (4 - ry) * ((4 - rx) * p0[2 * e] + rx * p0[2 * e + 1])
              + ry * ((4 - rx) * p1[2 * e] + rx * p1[2 * e + 1]);
          val[r][e] = (v + 8) >> 4;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++)
          val[r][e] = fetch_ref < PC > (refs[r], job.ref_stride[r], job.w, job.h,
              fx[r] + (4 * seg + e) * (1 << prec), fy[r] + row * (1 << prec), prec, ps, cb);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++)
      pred[e] = mode == 3 ? (val[0][e] + val[1][e] + 1) >> 1 : (mode == 1 ? val[0][e] : val[1][e]);
  }
  int wy = s_wy[row];
  if (y < job.yoff)
    wy += s_wy[2 * job.yoff - row - 1];
  if (y >= yfold_hi)
    wy += s_wy[2 * (job.yblen - job.yoff) - row - 1];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int x = xs + e, idx = 4 * seg + e;
    if (idx >= job.xblen)
      continue;
    int wx = s_wx[idx];
    if (x < job.xoff)
      wx += s_wx[2 * job.xoff - idx - 1];
    if (x >= xfold_hi)
      wx += s_wx[2 * (job.xblen - job.xoff) - idx - 1];
    int half;
    uint32_t *aw = acc_word < G, UV > (acc, par, x - x_lo, y - y_lo, &half);
    if constexpr (UV)
      half = cb;
    const uint32_t v = (uint32_t) (pred[e] * wx * wy);
    if (exact)
      acc_add_exact (aw, half, v & 0xffffu);
    else
      atomicAdd (aw, (v & 0xffffu) << (16 * half));
  }
}

// out = sat_u8 (residual + ((acc + 32) >> 6)) for one tile
template < typename G >
__device__ __forceinline__ bool
row_finish_is_fast (const ObmcJob & job, const PlaneIO & io, int x_lo, int x_hi)
{
  return job.res_bpp == 2 && x_hi - x_lo == G::kTW
      && ((((uintptr_t) io.residual) | (uintptr_t) io.residual_stride) & 15) == 0
      && ((((uintptr_t) io.out) | (uintptr_t) io.out_stride) & 7) == 0;
}

// the fast finish's residual: 8 pixels of one row per lane and round (UV: of both planes), fetched before the
// tile's last barrier
template < int TH, typename G > constexpr int kRFinishRounds = (TH * (G::kTW / 8) + kRThreads - 1) / kRThreads;

// The residual is read once and the picture written once; the half-pel planes are gathered from by
// every picture of the batch.  Streaming (non-temporal) accesses for the first two leave the caches
// to the planes: OBMC 0.2716 -> 0.2509 ms per 8 x 2160p step, the step 0.428 -> 0.410 (loads alone:
// 0.257 / 0.417).  Half the distinct reference bytes are worth 7 % of OBMC (bench.py
// SCHRO_BENCH_ONE_REF): it is the planes' residency that pays.
// (res[STEP * n + FIRST]: UV keeps the two planes' pieces of a round side by side)
template < int TH, typename G, int STEP, int FIRST >
__device__ __forceinline__ void
row_finish_prefetch (const PlaneIO & io, int tid, int x_lo, int y_lo, int y_hi, u32x4 * res)
{
  constexpr int kG = G::kTW / 8;        // 8-pixel groups per tile row (a power of two)
#pragma unroll
  for (int n = 0; n < kRFinishRounds < TH, G >; n++) {
    const int it = tid + n * kRThreads, g = it & (kG - 1), y = y_lo + it / kG;
    if (!io.residual)           // no residual to add (a zero_residual picture, schrodecoder.c:1904-1906)
      res[STEP * n + FIRST] = (u32x4) { 0u, 0u, 0u, 0u };
    else if (y < y_hi && it < TH * kG)
      res[STEP * n + FIRST] = __builtin_nontemporal_load ((const SCHRO_GLOBAL u32x4 *) ((const char *) io.residual + (size_t) y * io.residual_stride + 2 * (x_lo + 8 * g)));
  }
}

// orc_rrshift6_add_s16_2d / _s32_2d on one pixel per lane and step (any residual depth, any alignment);
// UV: component cb of the pixels' words
template < int TH, typename G, bool UV >
__device__ __forceinline__ void
row_finish_plain (const ObmcJob & job, const PlaneIO & io, int cb, const uint32_t * acc, int par, int tid, int x_lo, int y_lo,
    int x_hi, int y_hi)
{
  for (int it = tid; it < TH * G::kTW; it += kRThreads) {
    const int xx = it & (G::kTW - 1), yy = it / G::kTW;
    const int x = x_lo + xx, y = y_lo + yy;
    if (y >= y_hi || x >= x_hi)
      continue;
    int half;
    const uint32_t *aw = acc_word < G, UV > (const_cast < uint32_t * >(acc), par, xx, yy, &half);
    if constexpr (UV)
      half = cb;
    const int16_t a = (int16_t) (*aw >> (16 * half));
    const char *rrow = (const char *) io.residual + (size_t) y * io.residual_stride;
    const int16_t res = !io.residual ? (int16_t) 0 : job.res_bpp == 2 ? gload < int16_t > ((const int16_t *) rrow + x)
        : (int16_t) gload < int32_t > ((const int32_t *) rrow + x);   // convlw
    int16_t t1 = (int16_t) (a + 32);
    t1 = (int16_t) (t1 >> 6);
    t1 = (int16_t) (res + t1);
    gstore < uint8_t > (io.out + (size_t) y * io.out_stride + x, (uint8_t) clampi (t1, 0, 255));
  }
}

// NOCLAMP (r05): a prediction alone whose tile holds no DC value outside 0 .. 255 -- the weights of the blocks over a
// pixel add up to 64 and every prediction sample is a byte, so (sum + 32) >> 6 is one too: no clamp
template < int TH, typename G, bool NOCLAMP = false >
__device__ __forceinline__ void
row_finish (const ObmcJob & job, const PlaneIO & io, uint32_t * acc, int par, int tid, int x_lo, int y_lo,
    int x_hi, int y_hi, bool fast, const u32x4 * res)
{
  constexpr int kG = G::kTW / 8;
  if (fast) {
    // one lane: 8 pixels of one row, packed 16-bit arithmetic (the reference's adds wrap at 16 bits)
#pragma unroll
    for (int n = 0; n < kRFinishRounds < TH, G >; n++) {
      const int it = tid + n * kRThreads;
      const int g = it & (kG - 1), yy = it / kG;
      const int y = y_lo + yy;
      if (y >= y_hi || it >= TH * kG)
        continue;
      const uint32_t *ap = acc + yy * G::kAccW + (G::kMargin / 2 + 4 * g);
      uint32_t av[4];
      if (par) {
        // pixel 8 g sits in the high half of word 4 g + 8: shift the five words down by one pixel
        const uint32_t w0 = ap[0], w1 = ap[1], w2 = ap[2], w3 = ap[3], w4 = ap[4];
        av[0] = __builtin_amdgcn_alignbit (w1, w0, 16);
        av[1] = __builtin_amdgcn_alignbit (w2, w1, 16);
        av[2] = __builtin_amdgcn_alignbit (w3, w2, 16);
        av[3] = __builtin_amdgcn_alignbit (w4, w3, 16);
      } else {
        av[0] = ap[0];          // (rows are an odd number of words apart: no 16-byte reads)
        av[1] = ap[1];
        av[2] = ap[2];
        av[3] = ap[3];
      }
      const int x = x_lo + 8 * g;
      const u32x4 r = res[n];
      const uint32_t rv[4] = { r.x, r.y, r.z, r.w };
      uint32_t t[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        s16x2 v = (__builtin_bit_cast (s16x2, av[k]) + (short) 32) >> 6;
        v = v + __builtin_bit_cast (s16x2, rv[k]);
        if constexpr (!NOCLAMP)
          v = __builtin_elementwise_min (__builtin_elementwise_max (v, (s16x2) (short) 0), (s16x2) (short) 255);
        t[k] = __builtin_bit_cast (uint32_t, v);
      }
      u32x2 o;
      o.x = __builtin_amdgcn_perm (t[1], t[0], 0x06040200u);
      o.y = __builtin_amdgcn_perm (t[3], t[2], 0x06040200u);
      __builtin_nontemporal_store (o, (SCHRO_GLOBAL u32x2 *) (io.out + (size_t) y * io.out_stride + x));
    }
    return;
  }
  row_finish_plain < TH, G, false > (job, io, 0, acc, par, tid, x_lo, y_lo, x_hi, y_hi);
}

// UV: a lane takes 8 pixels of one row of BOTH planes: eight accumulator words (U sum low, V sum high),
// 16 bytes of each plane's residual (res[2 n], res[2 n + 1]), 8 bytes of each plane's picture
template < int TH, typename G, bool NOCLAMP = false >
__device__ __forceinline__ void
row_finish_uv (const PlaneIO & iou, const PlaneIO & iov, const uint32_t * acc, int tid, int x_lo, int y_lo, int y_hi, const u32x4 * res)
{
  constexpr int kG = G::kTW / 8;
#pragma unroll
  for (int n = 0; n < kRFinishRounds < TH, G >; n++) {
    const int it = tid + n * kRThreads;
    const int g = it & (kG - 1), yy = it / kG;
    const int y = y_lo + yy;
    if (y >= y_hi || it >= TH * kG)
      continue;
    const uint32_t *ap = acc + yy * G::kAccW + (G::kMargin + 8 * g);
    const u32x4 ru = res[2 * n], rv = res[2 * n + 1];
    const uint32_t ruw[4] = { ru.x, ru.y, ru.z, ru.w }, rvw[4] = { rv.x, rv.y, rv.z, rv.w };
    uint32_t t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      // the pixel's residuals as (U, V): the low / high 16 bits of the planes' dwords
      const uint32_t r = __builtin_amdgcn_perm (rvw[k >> 1], ruw[k >> 1], (k & 1) ? 0x07060302u : 0x05040100u);
      s16x2 v = (__builtin_bit_cast (s16x2, ap[k]) + (short) 32) >> 6;
      v = v + __builtin_bit_cast (s16x2, r);
      if constexpr (!NOCLAMP)
        v = __builtin_elementwise_min (__builtin_elementwise_max (v, (s16x2) (short) 0), (s16x2) (short) 255);
      t[k] = __builtin_bit_cast (uint32_t, v);
    }
    // t[k] = (U_k, 0, V_k, 0) -> (U0 U1 V0 V1), (U2 U3 V2 V3) -> U0 .. U3 | V0 .. V3
    const uint32_t a01 = __builtin_amdgcn_perm (t[1], t[0], 0x06020400u), a23 = __builtin_amdgcn_perm (t[3], t[2], 0x06020400u);
    const uint32_t a45 = __builtin_amdgcn_perm (t[5], t[4], 0x06020400u), a67 = __builtin_amdgcn_perm (t[7], t[6], 0x06020400u);
    u32x2 ou, ov;
    ou.x = __builtin_amdgcn_perm (a23, a01, 0x05040100u);
    ou.y = __builtin_amdgcn_perm (a67, a45, 0x05040100u);
    ov.x = __builtin_amdgcn_perm (a23, a01, 0x07060302u);
    ov.y = __builtin_amdgcn_perm (a67, a45, 0x07060302u);
    const int x = x_lo + 8 * g;
    __builtin_nontemporal_store (ou, (SCHRO_GLOBAL u32x2 *) (iou.out + (size_t) y * iou.out_stride + x));
    __builtin_nontemporal_store (ov, (SCHRO_GLOBAL u32x2 *) (iov.out + (size_t) y * iov.out_stride + x));
  }
}

// NORES (r05): every job of the launch is a prediction_only job (no residual to add: the combine form's launches,
// bench.py's headline).  The eight registers that carry the prefetched residual through the passes are not held at all.
// RK, NS (r06): the reference kind and the segments of a block row, see the head of the file.
// WP (r06): the picture's weights are not 1, 1 / 2 (a fade: blend_weighted).
// PAD (r06): the texture path takes a wave's load four lanes at a time and spends a cycle per 128-byte LINE the four touch
// (scripts/ta_rate_bench.hip: 16 / 32 / 64 cycles for lanes 32 / 64 / 128 bytes apart, and the same 64 addresses dealt so that
// line mates are 4 lanes apart cost 64 where neighbours cost 16).  A line of the tiled planes holds 4 rows: the items of a block are
// laid out so that every quad of lanes is the four rows of ONE line of the block's (first) reference -- the run starts on a
// multiple of 4 lanes, rows in front of / behind the window's in the same line are lanes that load a neighbour's row and add nothing.
constexpr int kRPadItems = 344; // the padding lanes a tile may spend (what its LDS share leaves); blocks beyond: edge class

template < int ND, int NP, bool UV = false, int TH = kRTH, bool NORES = false, int RK = 1, int NS = 1, bool WP = false, bool PAD = false >
__device__ __forceinline__ void
obmc_row_body (const ObmcJob * __restrict__ jobs, int njobs, const uint32_t * __restrict__ order, uint32_t * __restrict__ overflow,
    const uint32_t * __restrict__ wtabs)
{
  typedef RowGeo < ND, UV, NS > G;
  static_assert (!UV || NP == 1, "a UV job is one virtual plane of (U, V) samples");
  static_assert (NS == 1 || (NS == 2 && (ND == 3 || ND == 4)), "segments: halves of 12 or 16 bytes");
  constexpr int kRTW = G::kTW, ps = UV ? 1 : 0;
  __shared__ __attribute__ ((aligned (16))) uint32_t acc[TH * G::kAccW + 3];
  // (r05: s_wp | s_wx | s_wy are ONE block, copied from the job's weight table, see below)
  static_assert (!PAD || (RK != 0 && NS == 1), "line-aligned quads: the tiled planes, one segment");
  constexpr int kRBlkCap = G::kBlk, kRItemCap = G::kItem + (PAD ? kRPadItems : 0);
  typedef RowBlkT < G::kPadBlk > RowBlk;
  __shared__ RowBlk s_hot[kRBlkCap];            // the tile's blocks, in raster order
  __shared__ uint16_t s_meta[kRBlkCap];         // slot | first item within the slot << 5
  __shared__ uint16_t s_rim[kRBlkCap];          // the picture-rim blocks
  __shared__ uint16_t s_item[kRItemCap];
  __shared__ __attribute__ ((aligned (16))) uint32_t s_wp[G::kWTab];   // (row, pair) weights + folded x pairs, folded y (edge class) + the ramps
  const int *const s_wx = reinterpret_cast < const int * >(s_wp + G::kWRampX), *const s_wy = reinterpret_cast < const int * >(s_wp + G::kWRampY);    // (obmc_row_form: blocks up to 16 NS x 32)
  __shared__ int s_icnt[kRSlots];               // items of each slot
  __shared__ int s_nrim, s_wide, s_padleft;

  const uint64_t t_start = __builtin_amdgcn_s_memtime ();
#if defined (SCHRO_HIP_EXPERIMENTS) && defined (SCHRO_ROW_PRIO_SETUP)
  __builtin_amdgcn_s_setprio (SCHRO_ROW_PRIO_SETUP);
#endif
  const int bid = xcd_tile_id (blockIdx.x, gridDim.x);
  // r05: a tile's record from the host's order table (plane_obmc.cpp: obmc_tile_order) -- four words: job << 16 | tile,
  // x_lo | y_lo << 16, first block column | first block row << 16, block columns | block rows << 8 | ceil (2^16 /
  // block columns) << 16 -- one scalar load where every wave worked the same five divisions out (~90 scalar
  // instructions and a dozen branches of a wave's 430).  The experiments build keeps the arithmetic for runs without
  // the table (SCHRO_HIP_OBMC_ORDER=0).
#ifdef SCHRO_HIP_EXPERIMENTS
  const bool have_rec = order != nullptr;
#else
  constexpr bool have_rec = true;
#endif
  const u32x4 rec = have_rec ? *reinterpret_cast < const u32x4 * >(order + 4 * (size_t) bid) : (u32x4) { 0u, 0u, 0u, 0u };   // (a uniform address: a scalar load)
  const uint32_t entry = rec.x;
  const int ji = have_rec ? (int) (entry >> 16) : find_job (jobs, njobs, bid);
  const ObmcJob job = jobs[ji];
  // scratch runs (SCHRO_HIP_OBMC_STAMPS): cycles since the workgroup started, per phase
  // (r05: experiments build only -- each stamp is a handful of scalar instructions and a branch in EVERY wave: OBMC per
  // 8 x 2160p step 0.1640 -> 0.1616 ms without them)
#ifdef SCHRO_HIP_EXPERIMENTS
#define RSTAMP(n) do { if (job.stamps && threadIdx.x == 0 && blockIdx.x < 16384) \
    job.stamps[blockIdx.x * 16 + (n)] = __builtin_amdgcn_s_memtime () - t_start; } while (0)
#else
#define RSTAMP(n) do { } while (0)
#endif
  RSTAMP (7);                   // (the job is here)
  const int tid = threadIdx.x;
  int x_lo = (int) (rec.y & 0xffffu), y_lo = (int) (rec.y >> 16);
  if (!have_rec) {
    const int t = bid - job.tile_base;
    const int ty = mdiv (t, job.tiles_x, job.m_tiles_x), tx = t - ty * job.tiles_x;
    x_lo = tx * kRTW;
    y_lo = ty * TH;
  }
  const int x_hi = min (x_lo + kRTW, job.w), y_hi = min (y_lo + TH, job.h);
  constexpr int nplanes = NP;   // (every job of a launch has NP planes: the host groups them so)

  // scratch builds (experiments only; scripts/build_variant_one.sh ... -DSCHRO_HIP_EXPERIMENTS -DSCHRO_ROW_REP_SETUP=2): what
  // a phase costs in LAUNCH time, as opposed to where a tile's lifetime goes -- everything in front of the passes (clear,
  // tables, vectors, decode, sort, items) or the passes themselves done N times over (the pictures are then wrong)
#if defined (SCHRO_HIP_EXPERIMENTS) && defined (SCHRO_ROW_REP_SETUP)
  constexpr int kRepSetup = SCHRO_ROW_REP_SETUP;
#else
  constexpr int kRepSetup = 1;
#endif
#if defined (SCHRO_HIP_EXPERIMENTS) && defined (SCHRO_ROW_REP_PASSES)
  constexpr int kRepPasses = SCHRO_ROW_REP_PASSES;
#else
  constexpr int kRepPasses = 1;
#endif
#if defined (SCHRO_HIP_EXPERIMENTS) && defined (SCHRO_ROW_REP_FINISH)
  constexpr int kRepFinish = SCHRO_ROW_REP_FINISH;
#else
  constexpr int kRepFinish = 1;
#endif
  const int xblen = job.xblen, yblen = job.yblen;
  const int par = UV ? 0 : job.xoff & 1;        // block origins xbsep * i - xoff are odd: shift the accumulator by a pixel
  const int xfold_hi = job.nbx * job.xbsep - job.xoff, yfold_hi = job.nby * job.ybsep - job.yoff;
  int ibase[kRRim + 1];         // first item of each class
  bool exact = false;           // a DC value outside 0..255 in the tile: 16-bit sums may wrap
  int nrim = 0;
  constexpr int kAccQuads = (TH * G::kAccW + 3) / 4;   // the accumulator tile is cleared 16 bytes at a time
#pragma unroll
  for (int rep_ = 0; rep_ < kRepSetup; rep_++) {
  if (rep_)
    __syncthreads ();
  for (int it = tid; it < kAccQuads; it += kRThreads)
    reinterpret_cast < u32x4 * >(acc)[it] = (u32x4) { 0u, 0u, 0u, 0u };
  // r05: the plane geometry's weights come ready-made from the host (the job's `ipw` names its table in `wtabs`,
  // obmc_row_weight_table below).  The two waves that do not decode ask for it here, 16 bytes per lane, and put it into
  // LDS behind the first barrier, beside the decode: no ramps through LDS and no table arithmetic (~120 branchy
  // instructions in two waves).  (Waited for in front of that barrier by every lane it was slower: 0.1655 against 0.1616.)
  // (r06: the tables of the two-segment forms take two quads per lane)
  constexpr int kWQuads = G::kWTab / 4, kWQ = (kWQuads + kRThreads / 2 - 1) / (kRThreads / 2);
  static_assert (G::kWTab % 4 == 0 && kWQ <= 2, "one or two quads of the weight table per lane of the last two waves");
  u32x4 wq[kWQ];
#pragma unroll
  for (int n = 0; n < kWQ; n++) {
    wq[n] = (u32x4) { 0u, 0u, 0u, 0u };
    const int q = tid - kRThreads / 2 + n * (kRThreads / 2);
    if (tid >= kRThreads / 2 && q < kWQuads)
      wq[n] = gload < u32x4 > (wtabs + (size_t) job.ipw * G::kWTab + 4 * q);
  }
  if (tid >= 128 && tid < 128 + kRSlots)
    s_icnt[tid - 128] = 0;
  if (tid == 192) {
    s_nrim = 0;
    s_wide = 0;
    s_padleft = kRPadItems;
  }

  int nblk;
  {
    // ---- decode: every block whose footprint meets the tile, one per thread and round -----
    const int xbsep = job.xbsep, ybsep = job.ybsep, xoff = job.xoff, yoff = job.yoff, prec = job.prec;
    // (obmc_row_tile_record below is the same arithmetic on the host)
    int i_lo = (int) (rec.z & 0xffffu), j_lo = (int) (rec.z >> 16), nbi = (int) (rec.w & 0xffu), nbj = (int) ((rec.w >> 8) & 0xffu);
    // (blk / nbi as (blk * ceil (2^16 / nbi)) >> 16: exact while blk * nbi < 2^16 -- here blk < kRBlkCap <= 344 and nbi <= 66)
    static_assert (kRBlkCap * 128 < 65536, "the 16-bit block-row division below");
    uint32_t m16_nbi = rec.w >> 16;
    if (!have_rec) {
      i_lo = max (0, mdiv (x_lo + xoff - xblen + 2 * xbsep, xbsep, job.m_xbsep) - 1);
      const int i_hi = min (job.nbx - 1, mdiv (x_hi - 1 + xoff, xbsep, job.m_xbsep));
      j_lo = max (0, mdiv (y_lo + yoff - yblen + 2 * ybsep, ybsep, job.m_ybsep) - 1);
      const int j_hi = min (job.nby - 1, mdiv (y_hi - 1 + yoff, ybsep, job.m_ybsep));
      nbi = i_hi - i_lo + 1;
      nbj = j_hi - j_lo + 1;
      m16_nbi = nbi > 1 ? (65536u + (uint32_t) nbi - 1u) / (uint32_t) nbi : 0u;
    }
    // (r06, NS > 1: the record's block columns count SEGMENTS -- obmc_row_tile_record -- and so does the division)
    if (!have_rec && NS > 1) {
      nbi *= NS;
      m16_nbi = (65536u + (uint32_t) nbi - 1u) / (uint32_t) nbi;
    }
    nblk = nbi > 0 && nbj > 0 ? min (nbi * nbj, kRBlkCap) : 0;   // (the host sends larger geometries to obmc.hip)
    const int gh = 2 * job.h - 2;       // last valid half-pel sample row
    const int seglen = NS > 1 ? xblen / NS : xblen;
    // the motion vectors of the first round start their way from memory beside the set-up
    uint32_t mv_pre[3] = { 0u, 0u, 0u };
    if (tid < nblk) {
      const int bj = nbi == 1 ? tid : (int) (((uint32_t) tid * m16_nbi) >> 16);
      const uint8_t *mvp = job.mvs + (size_t) 20 * ((size_t) (j_lo + bj) * job.nbx + (i_lo + (tid - bj * nbi) / NS));
      mv_pre[0] = gload < uint32_t > (mvp);
      mv_pre[1] = gload < uint32_t > (mvp + 12);
      mv_pre[2] = gload < uint32_t > (mvp + 16);
    }
    RSTAMP (10);                // (accumulator cleared, ramps, the vectors asked for)
    // (r05, measured slower: the vectors asked for first and the weight tables made from weight_1d in front of this ONE
    // barrier, the ramps' round trip through LDS and the phase behind the barrier gone -- 0.1655 against 0.1611 ms per step)
    __syncthreads ();           // counters, the accumulator
    RSTAMP (8);
#pragma unroll
    for (int n = 0; n < kWQ; n++) {
      const int q = tid - kRThreads / 2 + n * (kRThreads / 2);
      if (tid >= kRThreads / 2 && q < kWQuads)
        reinterpret_cast < u32x4 * >(s_wp)[q] = wq[n];  // (read in the passes, two barriers on)
    }
    RSTAMP (1);
    for (int blk = tid; blk < nblk; blk += kRThreads) {
      const int bj = nbi == 1 ? blk : (int) (((uint32_t) blk * m16_nbi) >> 16);
      const int vi = blk - bj * nbi, seg = NS > 1 ? vi % NS : 0;
      const int i = i_lo + vi / NS, jj = j_lo + bj;
      uint32_t flags = mv_pre[0], v01 = mv_pre[1], v23 = mv_pre[2];
      if (blk >= kRThreads) {
        const uint8_t *mvp = job.mvs + (size_t) 20 * ((size_t) jj * job.nbx + i);
        flags = gload < uint32_t > (mvp);
        v01 = gload < uint32_t > (mvp + 12);
        v23 = gload < uint32_t > (mvp + 16);
      }
      const int bx = xbsep * i - xoff, by = ybsep * jj - yoff;
      RowBlk info;
      info.x = (int16_t) (bx + seg * seglen - x_lo);    // (the record stands for segment `seg` of the block's rows)
      info.y = (int16_t) (by - y_lo);
      const int mode = flags & 3;
      const bool interior = i >= 1 && i < job.max_x_blocks && jj >= 1 && jj < job.max_y_blocks;
      auto dc_of = [&] (int comp) {
        const int dc = comp == 0 ? (int16_t) (v01 & 0xffff) : comp == 1 ? (int16_t) (v01 >> 16) : (int16_t) (v23 & 0xffff);
        // get_dc_block stores a uint8_t; block_acc_dc multiplies a 16-bit parameter
        return interior ? (int) (int16_t) (dc + 128) : (int) (uint8_t) (dc + 128);
      };
      const int pdc = dc_of (job.comp), pdc_b = (nplanes > 1 || UV) ? dc_of (job.comp_b) : 0;
      const int dcs = (int) (((uint32_t) pdc & 0xffffu) | ((uint32_t) pdc_b << 16));
      uint32_t bflags = (uint32_t) mode | ((uint32_t) seg << 8);
      int taps[2] = { 0, 0 };
      RowRef in_ref[2], edge_ref[2];    // the window as the row classes / the edge class address it
      bool off_h = false, clamped_v = false;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        int fx, fy;
        mv_origin (job, bx, by, v01, v23, r, &fx, &fy);
        fx += (seg * seglen) * (1 << prec);
        const bool used = (mode & (r + 1)) != 0;
        if constexpr (RK == 0) {
          // plain planes (mv_precision 0): a window inside the picture is one run per row; one that leaves it
          // (get_block lets it, by up to 32 pixels) is clamped row by row and dword by dword in the edge class
          const bool in_h = fx >= 0 && fx + seglen <= job.w, in_v = fy >= 0 && fy + yblen <= job.h;
          clamped_v |= used && !(in_h && in_v);
          in_ref[r].base = used && in_h && in_v ? fx : 0;
          in_ref[r].ydb = used && in_h && in_v ? (uint32_t) fy : 0u;
          in_ref[r].dci = 0;
          edge_ref[r].base = used ? fx : 0;
          edge_ref[r].ydb = used ? (uint32_t) fy & 0xffffu : 0u;
          edge_ref[r].dci = 0;
        } else {
          // half-pel origin and the position inside the half-pel cell (prec 1: half-pel units; prec 2: quarter-pel
          // units, RK 3: eighth-pel units -- rx, ry 0 .. 3)
          const int hx = RK == 3 ? fx >> 2 : prec >= 2 ? fx >> 1 : fx, hy = RK == 3 ? fy >> 2 : prec >= 2 ? fy >> 1 : fy;
          const int rx = RK == 3 ? fx & 3 : prec >= 2 ? fx & 1 : 0, ry = RK == 3 ? fy & 3 : prec >= 2 ? fy & 1 : 0;
          const int wbits = RK == 3 ? (rx << 4) | (ry << 6) : 0;
          // columns: get_block's clamp keeps every window inside the aprons (32 pixels either side); the
          // test is a guard, not a case.  xp: the window's first byte column (UV: samples of two bytes)
          const int xp = ((hx >> 1) + kHpApron) << ps, px = hx & 1, py = hy & 1;
          const bool in_h = xp >= 0 && (xp >> 4) < (job.ref_stride[r] >> 9) && (unsigned) (hy + 16384) < 32768u;
          // rows: both taps of every sample row of the block inside the image, else clamped row by row
          const bool in_v = hy >= 0 && hy + 2 * (yblen - 1) + 1 <= gh;
          off_h |= used && !in_h;
          clamped_v |= used && !in_v;
          const int colbase = in_h && used ? (xp >> 4) * 512 + (xp & 15) + px * 128 : 0;
          const uint32_t dB = in_h && used && rx ? (uint32_t) (px ? (1 << ps) - 128 : 128) : 0u;
          in_ref[r].base = colbase + (in_h && used ? py * 256 : 0);
          in_ref[r].ydb = (in_h && in_v && used ? (uint32_t) (hy >> 1) : 0u) | (dB << 16);
          in_ref[r].dci = in_h && used ? ((ry ? (py ? (int) (((uint32_t) -256 << 16) | 1u) : (int) (256u << 16)) : 0) | wbits) : 0;
          edge_ref[r].base = colbase;
          edge_ref[r].ydb = (in_h && used ? (uint32_t) hy & 0xffffu : 0u) | (dB << 16);
          edge_ref[r].dci = in_h && used ? wbits : 0;
          bflags |= (uint32_t) (in_h && used && ry ? 1 : 0) << (6 + r);
          taps[r] = in_h && used ? (rx ? 1 : 0) | (ry ? 2 : 0) : 0;
        }
      }
      info.r[0] = in_ref[0];
      info.r[1] = in_ref[1];
      const int ra = max (0, -(int) info.y), rb = min (yblen, y_hi - by);
      // weights fold where the block hangs over the picture's rim: top | bottom << 1 | left << 2 | right << 3
      // (a segment folds at the side it lies on: the folded columns are the first 2 xoff and those from xbsep on)
      int fold = (by < yoff ? 1 : 0) | (by + yblen > yfold_hi ? 2 : 0) | (bx < xoff ? 4 : 0) | (bx + xblen > xfold_hi ? 8 : 0);
      if (NS > 1)
        fold &= seg == 0 ? ~8 : ~4;
      const bool wide_dc = mode == 0 && ((unsigned) pdc > 255u || (unsigned) pdc_b > 255u);
      if (wide_dc) {
        s_wide = 1;
        // (prediction_only launches, r04: such a prediction does not fit the u8 plane it is written to)
        if (overflow)
          __hip_atomic_store (overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      int key, nrows = rb - ra;
      if (off_h || wide_dc || yblen > 32 || xblen > NS * (UV ? 8 : 16)) {
        key = kRRim;
        // the rim path works on the WHOLE block (its first segment's record stands for it) from the clamped fetch
        // origins (16 bits each: obmc_row_form keeps planes whose origins do not fit away from this kernel)
        if (seg == 0) {
          int fx, fy;
          mv_origin (job, bx, by, v01, v23, 0, &fx, &fy);
          info.r[0].base = (int) (((uint32_t) fx & 0xffffu) | ((uint32_t) fy << 16));
          mv_origin (job, bx, by, v01, v23, 1, &fx, &fy);
          info.r[1].base = (int) (((uint32_t) fx & 0xffffu) | ((uint32_t) fy << 16));
          s_rim[atomicAdd (&s_nrim, 1)] = (uint16_t) blk;
        }
      } else if (fold || clamped_v) {
        // still a row per lane: sample rows clamped one by one, weights from the folded tables
        key = kREdge;
        info.r[0] = edge_ref[0];
        info.r[1] = edge_ref[1];
        bflags |= (uint32_t) fold << 2;
      } else if (mode == 3) {
        key = kRBoth;
      } else if (mode == 0) {
        key = kRDc;
      } else {
        key = mode == 1 ? kRRef0 : kRRef1;
      }
      // (NS > 1: a segment that lies beside the tile has no rows in it)
      if (NS > 1 && ((int) info.x + seglen <= 0 || (int) info.x >= kRTW))
        nrows = 0;
      // PAD: the block's run of lanes = whole lines of its (first) reference; a tile that has spent its padding sends the
      // block to the edge class (last in the item order: the runs in front of it stay on multiples of 4)
      int nitems = nrows;
      if constexpr (PAD) {
        if ((key == kRBoth || key == kRRef0 || key == kRRef1) && nrows > 0) {
          const uint32_t y_first = (info.r[key == kRRef1 ? 1 : 0].ydb & 0xffffu) + (uint32_t) ra;
          const int front = (int) (y_first & 3u), padded = (front + nrows + 3) & ~3, pad = padded - nrows;
          if (pad && atomicSub (&s_padleft, pad) < pad) {
            key = kREdge;
            info.r[0] = edge_ref[0];
            info.r[1] = edge_ref[1];
          } else {
            bflags |= (uint32_t) front << 9;
            nitems = padded;
          }
        }
      }
      // which taps the windows need: the slot inside the class
      const int slot = row_slot_base (key) + (key == kRBoth ? taps[0] | (taps[1] << 2) : key == kRRef0 ? taps[0] : key == kRRef1 ? taps[1] : 0);
      if (mode == 0)
        info.r[0].base = dcs;   // (no window: the field is free)
      info.fr = (uint32_t) ra | ((uint32_t) nrows << 8) | (bflags << 16);
      s_hot[blk] = info;
      // the block's rows take the next free items of its class (any order within a class will do)
      const int istart = key == kRRim ? 0 : atomicAdd (&s_icnt[slot], nitems);
      s_meta[blk] = (uint16_t) (slot | (istart << 5));
    }
  }
  __syncthreads ();
  RSTAMP (2);
  // first item of each slot: lane l of every wave scans the counters (no further barrier)
  int sbase;
  {
    const int lane = tid & 63;
    const int cnt = lane < kRSlots ? s_icnt[lane] : 0;
    // (r05: the scan in the lanes' own data path -- row_shr 1, 2, 4, 8 inside rows of 16 lanes, then lane 15 broadcast
    // into the row behind it: five additions where five __shfl_up were five LDS round trips and 45 instructions, in
    // EVERY wave of a launch that is bound by its vector instructions)
    static_assert (kRSlots <= 32, "the scan covers two rows of 16 lanes");
    int incl = cnt;
    incl += __builtin_amdgcn_update_dpp (0, incl, 0x111, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp (0, incl, 0x112, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp (0, incl, 0x114, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp (0, incl, 0x118, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp (0, incl, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    sbase = incl - cnt;
  }
#pragma unroll
  for (int c = 0; c <= kRRim; c++)
    ibase[c] = __builtin_amdgcn_readlane (sbase, row_slot_base (c));
  for (int b0 = 0; b0 < nblk; b0 += kRThreads) {       // (every lane takes part in the shuffle)
    const int blk = b0 + tid;
    const bool have = blk < nblk;
    const int meta = have ? s_meta[blk] : 0, slot = meta & 31;
    const int ib = __shfl (sbase, slot);
    if (!have || slot == row_slot_base (kRRim))
      continue;
    const int rows = (int) s_hot[blk].fr, ra = rows & 0xff, n = (rows >> 8) & 0xff;
    uint16_t *ip = s_item + ib + (meta >> 5);
    // (r05: four items per write -- the compiler's own vectorisation of the plain loop cost 85 instructions for 12 rows;
    // LDS takes the two-byte-aligned words as they come)
    typedef uint32_t u32_a2 __attribute__ ((aligned (2), may_alias));
    typedef u32x2 u32x2_a2 __attribute__ ((aligned (2), may_alias));
    if constexpr (PAD) {
      if (slot < row_slot_base (kRDc) && n > 0) {
        // whole quads: lane j of the run stands for row ra + j - front of the block; outside [ra, ra + n) it loads the nearest
        // row inside (the same line) and is marked
        const int front = (rows >> 25) & 3, total = (front + n + 3) & ~3;
#pragma clang loop vectorize(disable) unroll(disable)
        for (int j = 0; j < total; j += 4) {
          uint32_t it4[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int idx = j + q - front;
            it4[q] = (uint32_t) (blk | ((ra + min (max (idx, 0), n - 1)) << 9)) | ((unsigned) idx >= (unsigned) n ? 0x8000u : 0u);
          }
          *reinterpret_cast < u32x2_a2 * >(ip + j) = (u32x2) { it4[0] | (it4[1] << 16), it4[2] | (it4[3] << 16) };
        }
        continue;
      }
    }
    uint32_t pair = (uint32_t) (blk | (ra << 9)) * 0x00010001u + 0x02000000u;      // rows ra, ra + 1
    int r = 0;
#pragma clang loop vectorize(disable) unroll(disable)
    for (; r + 4 <= n; r += 4, pair += 0x08000800u)
      *reinterpret_cast < u32x2_a2 * >(ip + r) = (u32x2) { pair, pair + 0x04000400u };
    if (n & 2) {
      *reinterpret_cast < u32_a2 * >(ip + r) = pair;
      r += 2;
      pair += 0x04000400u;
    }
    if (n & 1)
      ip[r] = (uint16_t) pair;
  }
  exact = __builtin_amdgcn_readfirstlane (s_wide) != 0;
  nrim = __builtin_amdgcn_readfirstlane (s_nrim);
  __syncthreads ();
  RSTAMP (3);
  }                             // (kRepSetup)

  // ---- per plane of the job: accumulate, finish -----------------------------------------------
#pragma unroll
  for (int pl = 0; pl < nplanes; pl++) {
    PlaneIO io;
    io.ref[0] = pl ? job.ref_b[0] : job.ref[0];
    io.ref[1] = pl ? job.ref_b[1] : job.ref[1];
    io.residual = pl ? job.residual_b : job.residual;
    io.out = pl ? job.out_b : job.out;
    io.residual_stride = pl ? job.residual_stride_b : job.residual_stride;
    io.out_stride = pl ? job.out_stride_b : job.out_stride;
    // UV: the V plane's residual and picture (the references are the pair images in io)
    PlaneIO iov = io;
    if constexpr (UV) {
      if constexpr (RK == 0) {  // plain planes: the V planes of the references (the rim path reads them by component)
        iov.ref[0] = job.ref_b[0];
        iov.ref[1] = job.ref_b[1];
      }
      iov.residual = job.residual_b;
      iov.out = job.out_b;
      iov.residual_stride = job.residual_stride_b;
      iov.out_stride = job.out_stride_b;
    }
    // The residual of the fast finish is asked for before the passes where the registers allow
    // (8 per lane, held through the passes): it streams from HBM, and fetched after the passes
    // its latency was the tile's to wait for.
    constexpr bool kEarlyRes = ND >= 3 && NP == 1 && !NORES;
    const bool fast = row_finish_is_fast < G > (job, io, x_lo, x_hi) && (!UV || row_finish_is_fast < G > (job, iov, x_lo, x_hi));
    constexpr int kRounds = kRFinishRounds < TH, G >;
    u32x4 res[UV ? 2 * kRounds : kRounds];
#define SCHRO_ROW_PREFETCH() do { \
      if constexpr (UV) { \
        row_finish_prefetch < TH, G, 2, 0 > (io, tid, x_lo, y_lo, y_hi, res); \
        row_finish_prefetch < TH, G, 2, 1 > (iov, tid, x_lo, y_lo, y_hi, res); \
      } else { \
        row_finish_prefetch < TH, G, 1, 0 > (io, tid, x_lo, y_lo, y_hi, res); \
      } } while (0)
    if constexpr (kEarlyRes) {
      if (fast)
        SCHRO_ROW_PREFETCH ();
      __builtin_amdgcn_sched_barrier (0);
    }
    int turn = 0;
    // the references as buffers of whole bands of 4 plane rows (schro_hip_internal.h); RK 0: the plane's rows, the last
    // one up to its last dword (obmc_row_form: strides and pointers are multiples of 4) -- a run's dwords beyond it
    // read as zero and carry no weight
    RowRefs refs;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      refs.stride[r] = (uint32_t) job.ref_stride[r];
      const uint32_t bytes = RK == 0 ? (uint32_t) job.ref_stride[r] * (uint32_t) (job.h - 1) + (((uint32_t) job.w + 3u) & ~3u)
          : (uint32_t) job.ref_stride[r] * (uint32_t) ((job.h + 3) >> 2);
      refs.rsrc[r] = __builtin_amdgcn_make_buffer_rsrc ((void *) io.ref[r], 0, (int) bytes, 0x00020000);
      refs.rsrc_b[r] = refs.rsrc[r];
      if constexpr (UV && RK == 0)
        refs.rsrc_b[r] = __builtin_amdgcn_make_buffer_rsrc ((void *) iov.ref[r], 0, (int) bytes, 0x00020000);
    }
#define SCHRO_ROW_CLASS(C) row_class < ND, UV, C, RK, NS, WP, PAD > (job, pl, refs, s_item, s_hot, s_wp, acc, par, \
    ibase[C], ibase[C + 1], exact, &turn)
#pragma unroll
    for (int rep_ = 0; rep_ < kRepPasses; rep_++) {
      SCHRO_ROW_CLASS (kRBoth);
      SCHRO_ROW_CLASS (kRRef0);
      SCHRO_ROW_CLASS (kRRef1);
      SCHRO_ROW_CLASS (kRDc);
      SCHRO_ROW_CLASS (kREdge);
    }
#undef SCHRO_ROW_CLASS
    RSTAMP (4);
    // picture-rim blocks: exact clamp / fold path
    if (nrim > 0) {
      const int nseg = (xblen + 3) >> 2, per_block = yblen * nseg;
      const uint32_t m_per_block = div_magic (per_block), m_nseg = div_magic (nseg);
      for (int item = tid; item < nrim * per_block; item += kRThreads) {
        const int b = mdiv (item, per_block, m_per_block);
        const int rem = item - b * per_block;
        const int r2 = mdiv (rem, nseg, m_nseg), s2 = rem - r2 * nseg;
        const RowBlk & hb = s_hot[s_rim[b]];
        const int bx = hb.x + x_lo, by = hb.y + y_lo;
        const int y = by + r2, xs = bx + 4 * s2;
        if (y < y_lo || y >= y_hi || xs + 3 < x_lo || xs >= x_hi)
          continue;
        const int fx[2] = { (int) (int16_t) hb.r[0].base, (int) (int16_t) hb.r[1].base }, fy[2] = { hb.r[0].base >> 16, hb.r[1].base >> 16 };
#define SCHRO_ROW_SLOW(cb, dcpl) do { \
          const int md = (int) (blk_flags (hb) & 3u) | (blk_dc (hb, dcpl) << 8);       /* (the DC part is read in mode 0 only) */ \
          if constexpr (RK == 0) \
            row_slow < 0, G, UV > (job, (UV && cb) ? iov : io, cb, bx, by, md, fx, fy, r2, s2, x_lo, y_lo, xfold_hi, yfold_hi, s_wx, s_wy, acc, par, exact); \
          else if (RK == 1 && job.prec == 1) \
            row_slow < 1, G, UV > (job, io, cb, bx, by, md, fx, fy, r2, s2, x_lo, y_lo, xfold_hi, yfold_hi, s_wx, s_wy, acc, par, exact); \
          else \
            row_slow < 2, G, UV > (job, io, cb, bx, by, md, fx, fy, r2, s2, x_lo, y_lo, xfold_hi, yfold_hi, s_wx, s_wy, acc, par, exact); \
        } while (0)
        if constexpr (UV) {
          SCHRO_ROW_SLOW (0, 0);
          SCHRO_ROW_SLOW (1, 1);
        } else {
          SCHRO_ROW_SLOW (0, pl);
        }
#undef SCHRO_ROW_SLOW
      }
    }
    RSTAMP (5);
    if constexpr (NORES) {
#pragma unroll
      for (int n = 0; n < (UV ? 2 * kRounds : kRounds); n++)
        res[n] = (u32x4) { 0u, 0u, 0u, 0u };
    } else if constexpr (!kEarlyRes) {
      if (fast)
        SCHRO_ROW_PREFETCH ();
    }
#undef SCHRO_ROW_PREFETCH
    __syncthreads ();
    RSTAMP (6);
#pragma unroll
    for (int rep_ = 0; rep_ < kRepFinish; rep_++)
    if constexpr (UV) {
      if (fast) {
        if (NORES && !exact)
          row_finish_uv < TH, G, NORES > (io, iov, acc, tid, x_lo, y_lo, y_hi, res);
        else
          row_finish_uv < TH, G > (io, iov, acc, tid, x_lo, y_lo, y_hi, res);
      } else {
        row_finish_plain < TH, G, true > (job, io, 0, acc, par, tid, x_lo, y_lo, x_hi, y_hi);
        row_finish_plain < TH, G, true > (job, iov, 1, acc, par, tid, x_lo, y_lo, x_hi, y_hi);
      }
    } else {
      if (NORES && !exact)
        row_finish < TH, G, NORES > (job, io, acc, par, tid, x_lo, y_lo, x_hi, y_hi, fast, res);
      else
        row_finish < TH, G > (job, io, acc, par, tid, x_lo, y_lo, x_hi, y_hi, fast, res);
    }
    if (pl + 1 < nplanes) {     // the job's next plane starts from a zero accumulator
      __syncthreads ();
      for (int it = tid; it < kAccQuads; it += kRThreads)
        reinterpret_cast < u32x4 * >(acc)[it] = (u32x4) { 0u, 0u, 0u, 0u };
      __syncthreads ();
    }
  }
  RSTAMP (9);
#ifdef SCHRO_HIP_EXPERIMENTS
  if (job.stamps && threadIdx.x == 0 && blockIdx.x < 16384) {  // absolute start / end, where it ran
    job.stamps[blockIdx.x * 16 + 12] = t_start;
    job.stamps[blockIdx.x * 16 + 13] = __builtin_amdgcn_s_memtime ();
    job.stamps[blockIdx.x * 16 + 14] = __builtin_amdgcn_s_getreg ((4 << 0) | (0 << 6) | (31 << 11)); // HW_ID
    job.stamps[blockIdx.x * 16 + 15] = __builtin_amdgcn_s_getreg ((20 << 0) | (0 << 6) | (31 << 11));        // XCC_ID
  }
#else
  (void) t_start;
#endif
#undef RSTAMP
}

#define SCHRO_ROW_KERNEL(name, waves, ...) \
__global__ __launch_bounds__ (kRThreads) __attribute__ ((amdgpu_waves_per_eu (waves, waves))) \
void name (const ObmcJob * __restrict__ jobs, int njobs, const uint32_t * __restrict__ order, uint32_t * __restrict__ overflow, \
    const uint32_t * __restrict__ wtabs) \
{ \
  obmc_row_body < __VA_ARGS__ > (jobs, njobs, order, overflow, wtabs); \
}

typedef void (*RowKernel) (const ObmcJob *, int, const uint32_t *, uint32_t *, const uint32_t *);

}                               // namespace
}                               // namespace schro
