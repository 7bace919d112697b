// lowdelay.hip -- VC-2 low-delay transform data on the device.
//
//   slice_kernel        one thread per slice: exp-Golomb unpack + dequantise into the
//                       interleaved coefficient frame (schrolowdelay.c:109-310)
//   slice_run_kernel    r03: the same for slices that divide the sub-bands evenly, a step per
//                       non-zero value (zero runs counted from the bit window)
//   dc_predict_kernel   DC prediction of an LL band (schrodecoder.c:3219-3277)
//   dc_skew_kernel      r03: the same for bands of whole 16-byte pieces, strips of 64 rows on
//                       separate CUs, a lane one sample behind its neighbour
//
// Slices are fixed-size and independent (their offsets follow from the slice number,
// schrolowdelay.c:607-631), the codes inside one are strictly serial: a lane walks its
// own slice, a wave holds 64 neighbouring slices of one slice row.  Bit-exact with the
// reference's three decoders (see include/schro_hip.h for which one a picture gets).

#include "schro_hip_internal.h"

#include <algorithm>
#include <cstdlib>

namespace schro {

// schro_table_quant / schro_table_offset_1_2 (schrotables.c), by the generating formula of
// the Dirac specification (13.3.1); tests pin all 61 entries against the reference's
struct QuantTables {
  uint32_t factor[61], offset[61];
};
constexpr QuantTables
make_quant_tables ()
{
  QuantTables t = { };
  for (int q = 0; q <= 60; q++) {
    const uint64_t base = (uint64_t) 1 << (q / 4);
    const uint64_t f = (q & 3) == 0 ? 4 * base : (q & 3) == 1 ? (503829 * base + 52958) / 105917
        : (q & 3) == 2 ? (665857 * base + 58854) / 117708 : (440253 * base + 32722) / 65444;
    t.factor[q] = (uint32_t) f;
    t.offset[q] = q == 0 ? 1u : q == 1 ? 2u : (uint32_t) ((f + 1) / 2);
  }
  return t;
}
static __device__ const QuantTables kQuant = make_quant_tables ();

// MSB-first bit reader over global memory; bits at or beyond `end` read as 1 (the guard bit
// of schro_unpack_init_with_data (..., 1), schrolowdelay.c:126).  The 64 lanes of a wave
// read 64 different slices, and the coefficients the kernel streams out keep pushing the
// slice bytes out of L2 (rocprofv3: with a dword fetched per 32 bits the kernel pulled 1.8 GB
// through FETCH_SIZE for 80 MB of slices): a lane therefore fetches its slice in aligned
// 16-byte pieces, one piece ahead of the piece it is reading, and hands the dwords out of
// the two pieces it holds.
struct BitReader {
  const uint32_t *base;         // 16-byte aligned
  uint32_t pos, end;            // bit positions from base[0]
  uint32_t last_piece;          // the piece that holds the last byte of the buffer
  uint32_t w0, w1, widx;        // big-endian dwords widx and widx + 1
  u32x4 res, nxt;               // pieces pidx and pidx + 1 as they lie in memory
  uint32_t pidx;

  __device__ __forceinline__ u32x4 fetch_piece (uint32_t pi) const
  {
    return gload < u32x4 > (base + 4u * min (pi, last_piece));
  }
  static __device__ __forceinline__ uint32_t pick (const u32x4 & v, uint32_t k)
  {
    return __builtin_bswap32 (k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w);
  }
  // dword j, which lies in piece pidx or pidx + 1; moves on to the next piece when it does
  __device__ __forceinline__ uint32_t take (uint32_t j)
  {
    if ((j >> 2) != pidx) {
      res = nxt;
      pidx++;
      // (keeps the load below the copy: scheduled above it, its result needs registers of its own
      // and the move into nxt -- with a wait for the load -- comes right behind it)
      asm volatile ("" : "+v" (res) : : "memory");
      nxt = fetch_piece (pidx + 1);
    }
    return pick (res, j & 3u);
  }
  __device__ __forceinline__ void start (uint32_t p)
  {
    pos = p;
    widx = p >> 5;
    pidx = widx >> 2;
    res = fetch_piece (pidx);
    nxt = fetch_piece (pidx + 1);
    w0 = pick (res, widx & 3u);
    w1 = take (widx + 1);
  }
  __device__ __forceinline__ void skip (uint32_t n)
  {
    pos += n;
    if ((pos >> 5) > widx + 1)
      start (pos);
  }
  // the next 32 bits
  __device__ __forceinline__ uint32_t peek ()
  {
    const uint32_t wi = pos >> 5;
    if (wi != widx) {           // a code is at most 32 bits here: one dword further
      w0 = w1;
      w1 = take (wi + 1);
      widx = wi;
    }
    uint32_t v = __funnelshift_l (w1, w0, pos & 31u);
    const uint32_t left = pos < end ? end - pos : 0u;
    if (left < 32u)
      v |= left ? 0xffffffffu >> left : 0xffffffffu;
    return v;
  }
  __device__ __forceinline__ uint32_t bit ()
  {
    const uint32_t v = peek () >> 31;
    pos++;
    return v;
  }
  __device__ __forceinline__ uint32_t bits (int n)      // 0 <= n <= 32
  {
    if (n == 0)
      return 0;
    const uint32_t v = peek () >> (32 - n);
    pos += n;
    return v;
  }
  // schro_unpack_decode_sint: interleaved exp-Golomb, "0 b" per data bit, "1", sign
  __device__ __forceinline__ int32_t sint ()
  {
    const uint32_t v = peek ();
    const uint32_t stop = v & 0xaaaaaaaau;      // first 1 at an even position ends the code
    const int k = __clz ((int) stop);           // 2 * count (32 if none)
    if (k <= 30) {
      const int c = k >> 1;
      if (c == 0) {
        pos += 1;
        return 0;
      }
      // the c data bits sit at the odd positions of the top 2c bits
      uint32_t t = (v >> (32 - k)) & 0x55555555u;
      t = (t | (t >> 1)) & 0x33333333u;
      t = (t | (t >> 2)) & 0x0f0f0f0fu;
      t = (t | (t >> 4)) & 0x00ff00ffu;
      t = (t | (t >> 8)) & 0x0000ffffu;
      const uint32_t mag = (1u << c) - 1u + t;  // never 0 here
      const uint32_t neg = (v >> (30 - k)) & 1u;
      pos += k + 2;
      return neg ? -(int32_t) mag : (int32_t) mag;
    }
    // longer than the window (|value| >= 65535): bit by bit, modulo 2^32
    uint32_t count = 0, value = 0;
    while (!bit ()) {
      count++;
      value = (value << 1) | bit ();
    }
    value += (count < 32u ? 1u << count : 0u) - 1u;
    if (value && bit ())
      value = 0u - value;
    return (int32_t) value;
  }
};

// schro_dequantise (schroutils.c:180-189): int arithmetic
__device__ __forceinline__ int32_t
dequant_int (int32_t q, uint32_t factor, uint32_t offset)
{
  if (q == 0)
    return 0;
  const uint32_t mag = q < 0 ? 0u - (uint32_t) q : (uint32_t) q;
  const int32_t r = (int32_t) (mag * factor + offset + 2u) >> 2;
  return q < 0 ? -r : r;
}

// orc_dequantise_var_s16_ip (schroorc.orc:1204-1217): 16 bits at every step
__device__ __forceinline__ int32_t
dequant_s16 (int32_t value, uint32_t factor, uint32_t offset)
{
  const int16_t q = (int16_t) value;
  const int16_t f = (int16_t) factor, o = (int16_t) (offset + 2u);
  const int16_t sign = q > 0 ? 1 : (q < 0 ? -1 : 0);
  const int16_t mag = (int16_t) (q < 0 ? -q : q);
  int16_t t = (int16_t) (mag * f);
  t = (int16_t) (t + o);
  t = (int16_t) (t >> 2);
  return (int16_t) (t * sign);
}

__device__ __forceinline__ int
subband_position (int index)
{                               // schroparams.c:355-368
  return index == 0 ? 0 : ((index - 1) / 3 + 1) * 4 - 3 + (index - 1) % 3;
}

// Staging of one sub-band row of a wave's 64 slices: lane L puts its bw values at
// L * (bw + 1) (odd pitch: no bank conflicts), then the wave copies the 64 * bw values out
// 16 bytes per lane -- neighbouring slices are neighbours in the row, so the copy is one
// contiguous run per slice row instead of 64 scattered 2- or 4-byte stores per code.
constexpr int kStageWords = 64 * 34;    // luma: bw <= 33; chroma (U and V): bw <= 16

template < typename T, int ARITH >
__global__ __launch_bounds__ (64)
void slice_kernel (const SliceJob * __restrict__ jobs, const SliceParams P)
{
  __shared__ int32_t stage[kStageWords];
  __shared__ uint64_t rowbase[2][64];
  constexpr int E = 16 / (int) sizeof (T);      // samples per 16-byte store
  const SliceJob job = jobs[blockIdx.y];
  const int lane = (int) threadIdx.x;
  const int nslices = P.nh * P.nv;
  const int s0 = blockIdx.x * 64, s = s0 + lane;
  const bool active = s < nslices;
  const int nvalid = min (64, nslices - s0);
  const int sc = active ? s : nslices - 1;      // idle lanes shadow the last slice, read guard bits
  const int sy = sc / P.nh, sx = sc - sy * P.nh;
  // offset of slice s: s whole slices plus one extra byte per wrap of the accumulator
  const uint32_t wraps = (uint32_t) (((uint64_t) sc * (uint32_t) P.remainder) / (uint32_t) P.denom);
  const uint32_t wraps1 = (uint32_t) (((uint64_t) (sc + 1) * (uint32_t) P.remainder) / (uint32_t) P.denom);
  const uint32_t offset = (uint32_t) sc * (uint32_t) P.n_bytes + wraps;
  const uint32_t slice_bytes = (uint32_t) P.n_bytes + (wraps1 - wraps);

  const uintptr_t addr = (uintptr_t) job.data;
  BitReader yb;
  yb.base = (const uint32_t *) (addr & ~(uintptr_t) 15);
  const uint32_t lead = 8u * (uint32_t) (addr & 15);
  const uint32_t buffer_end = lead + 8u * job.data_bytes;
  yb.last_piece = (buffer_end - 1u) >> 7;
  yb.end = active ? lead + 8u * (offset + slice_bytes) : 0u;
  yb.start (lead + 8u * offset);
  const int base_index = (int) yb.bits (7);
  const uint32_t field = 8u * (ARITH == SCHRO_HIP_LOWDELAY_FAST16 ? (uint32_t) P.n_bytes : slice_bytes);
  const int length_bits = field ? 32 - __clz ((int) field) : 0;         // ilog2up, :94-105
  const uint32_t slice_y_length = yb.bits (length_bits);
  BitReader uvb = yb;
  yb.end = active ? min (yb.pos + slice_y_length, buffer_end) : 0u;     // schro_unpack_limit_bits_remaining
  uvb.skip (slice_y_length);    // schro_unpack_skip_bits

  const bool aligned = (((uintptr_t) job.comp[0] | (uintptr_t) job.comp[1] | (uintptr_t) job.comp[2]
          | (uintptr_t) job.stride[0] | (uintptr_t) job.stride[1] | (uintptr_t) job.stride[2]) & 15) == 0;
  const int nsub = 1 + 3 * P.depth;
  BitReader b = yb;             // (one reader in registers: a reference to either would live in scratch)
  // The luma codes and the (interleaved) chroma codes of a slice are two bit strings whose start
  // positions the slice header gives: with two workgroups per 64 slices (gridDim.z == 2) one
  // decodes the luma strings and the other the chroma strings -- twice the waves, half the serial
  // chain per lane (256 instead of 512 codes for 32x8 slices of 4:2:2).
  const int k_first = gridDim.z == 2 ? (int) blockIdx.z : 0, k_last = gridDim.z == 2 ? (int) blockIdx.z : 1;
#pragma unroll 1
  for (int k = k_first; k <= k_last; k++) {
    if (k)
      b = uvb;
    const int iwt_w = k ? P.iwt_cw : P.iwt_lw, iwt_h = k ? P.iwt_ch : P.iwt_lh;
    // luma alone, then U and V together (selected by value: indexing job.comp[] with k would
    // put the job into scratch)
    uint8_t *const plane[2] = { (uint8_t *) (k ? job.comp[1] : job.comp[0]), (uint8_t *) job.comp[2] };
    const int plane_stride[2] = { k ? job.stride[1] : job.stride[0], job.stride[2] };
#pragma unroll 1
    for (int i = 0; i < nsub; i++) {
      const int qi = min (max (base_index - P.quant_matrix[i], 0), 60);
      const uint32_t qf = kQuant.factor[qi], qo = kQuant.offset[qi];
      // schro_subband_get_frame_data + schro_frame_data_get_codeblock
      const int position = subband_position (i);
      const int shift = P.depth - (position >> 2);
      const int w = iwt_w >> shift, h = iwt_h >> shift;
      const int ebw = w / P.nh, ebh = h / P.nv;         // slice rectangle if the band divides evenly
      const bool even = ebw * P.nh == w && ebh * P.nv == h;
      const int xmin = even ? ebw * sx : (w * sx) / P.nh, xmax = even ? xmin + ebw : (w * (sx + 1)) / P.nh;
      const int ymin = even ? ebh * sy : (h * sy) / P.nv, ymax = even ? ymin + ebh : (h * (sy + 1)) / P.nv;
      const int bw = xmax - xmin;
      uint8_t *dst[2];
      size_t pitch[2];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        pitch[c] = (size_t) plane_stride[c] << shift;      // luma: c == 1 unused
        dst[c] = plane[c] + ((position & 2) ? pitch[c] >> 1 : 0)
            + ((position & 1) ? (size_t) w * sizeof (T) : 0) + (size_t) ymin * pitch[c] + (size_t) xmin * sizeof (T);
      }
      const int lpitch = ebw + 1;
      if (even && aligned && ebw % E == 0 && (k + 1) * 64 * lpitch <= kStageWords) {
        // ---- every lane has the same ebw x ebh rectangle: rows through LDS -------------
        rowbase[0][lane] = (uint64_t) (uintptr_t) dst[0];
        rowbase[1][lane] = (uint64_t) (uintptr_t) dst[1];
        const int cpl = ebw / E;        // 16-byte chunks per lane and row
        const uint32_t m_cpl = div_magic (cpl);
        const int nchunk = nvalid * cpl;
        int32_t *mine = stage + lane * lpitch;
        size_t yoff[2] = { 0, 0 };
#pragma unroll 1
        for (int y = 0; y < ebh; y++) {
#pragma unroll 1
          for (int x = 0; x < ebw; x++) {
            const int32_t v0 = b.sint ();
            mine[x] = ARITH == SCHRO_HIP_LOWDELAY_FAST16 ? dequant_s16 (v0, qf, qo) : dequant_int (v0, qf, qo);
            if (k) {            // U and V values alternate
              const int32_t v1 = b.sint ();
              mine[64 * lpitch + x] = ARITH == SCHRO_HIP_LOWDELAY_FAST16 ? dequant_s16 (v1, qf, qo) : dequant_int (v1, qf, qo);
            }
          }
          __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier ();
          __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
          for (int c = 0; c <= k; c++) {
#pragma unroll 1
            for (int ch = lane; ch < nchunk; ch += 64) {
              const int src = mdiv (ch, cpl, m_cpl), xc = ch - src * cpl;
              const int32_t *from = stage + c * 64 * lpitch + src * lpitch + xc * E;
              uint8_t *to = (uint8_t *) (uintptr_t) rowbase[c][src] + yoff[c] + (size_t) xc * 16;
              u32x4 o;
              if constexpr (E == 4) {
                o = u32x4 { (uint32_t) from[0], (uint32_t) from[1], (uint32_t) from[2], (uint32_t) from[3] };
              } else {
                o = u32x4 { ((uint32_t) from[0] & 0xffffu) | ((uint32_t) from[1] << 16),
                  ((uint32_t) from[2] & 0xffffu) | ((uint32_t) from[3] << 16),
                  ((uint32_t) from[4] & 0xffffu) | ((uint32_t) from[5] << 16),
                  ((uint32_t) from[6] & 0xffffu) | ((uint32_t) from[7] << 16) };
              }
              // streamed: written once, read by a later kernel; a normal store would push the
              // slice bytes this wave still has to read out of L2
              __builtin_nontemporal_store (o, (SCHRO_GLOBAL u32x4 *) to);
            }
          }
          __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier ();     // the next row overwrites the staging
          __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
          yoff[0] += pitch[0];
          yoff[1] += pitch[1];
        }
        continue;
      }
      // ---- ragged rectangles, odd widths, unaligned planes: store as decoded -----------
#pragma unroll 1
      for (int y = ymin; y < ymax; y++) {
#pragma unroll 1
        for (int x = 0; x < bw; x++) {
          const int32_t v0 = b.sint ();
          const int32_t d0 = ARITH == SCHRO_HIP_LOWDELAY_FAST16 ? dequant_s16 (v0, qf, qo) : dequant_int (v0, qf, qo);
          if (active)
            gstore < T > ((T *) dst[0] + x, (T) d0);
          if (k) {              // U and V values alternate
            const int32_t v1 = b.sint ();
            const int32_t d1 = ARITH == SCHRO_HIP_LOWDELAY_FAST16 ? dequant_s16 (v1, qf, qo) : dequant_int (v1, qf, qo);
            if (active)
              gstore < T > ((T *) dst[1] + x, (T) d1);
          }
        }
        dst[0] += pitch[0];
        dst[1] += pitch[1];
      }
    }
  }
}

// ---- r03: slice_run_kernel -- the same slices, a step per NON-ZERO value --------------------
// slice_kernel is bound by vector issue (rocprofv3: 75 VALU instructions per code and lane, the
// SIMDs issuing 80 % of the time), and most codes of a low-delay slice are the single bit "1" =
// value 0 (a 32x8 slice of config 5 has 2.4 bits per value to spend).  Here a lane's step is: count
// the leading ones of its 32-bit window (= that many zeros, which are already in the staging:
// it is zero-filled and every copy-out puts zeros back), then decode the code behind them out of
// the SAME window when it ends inside it.  A wave takes as many steps per staging turn as its
// busiest lane has non-zero values (+ one per 32 zeros in a row), not as many as there are values.
// A staging turn = as many whole rows of a sub-band's slice rectangle as fit kRunCap words per lane
// (U and V interleaved as they come); the copy-out is slice_kernel's (one contiguous run per slice
// row, 16 bytes per lane), rectangles narrower than 16 bytes go out value by value.
// The host takes this kernel when every sub-band divides evenly into the slices, a rectangle row
// fits the staging and the planes are 16-byte aligned (launch_slices); slice_kernel stays for the rest.
// Words per lane: the smallest of 16 / 32 / 64 that holds a rectangle row (16: 4.3 KB of LDS per
// wave; with 62 VGPRs 6 waves per SIMD are resident -- measured on config 5: 16 words 0.082 ms per
// picture, 32 words 0.090, 64 words 0.113: occupancy is worth more than longer turns).
constexpr int kRunCapMin = 16, kRunCapMax = 64;

// MSB-first reader whose dwords get their guard ones (bits at or beyond `end` read as 1) when they
// are fetched, not at every peek; between two peeks the position moves by at most 32 bits.
struct RunReader {
  const uint32_t *base;
  uint32_t pos, end, last_piece;
  uint32_t w0, w1, widx;
  u32x4 res, nxt;
  uint32_t pidx;

  __device__ __forceinline__ u32x4 fetch_piece (uint32_t pi) const
  {
    return gload < u32x4 > (base + 4u * min (pi, last_piece));
  }
  __device__ __forceinline__ uint32_t guarded (uint32_t w, uint32_t j) const
  {
    const uint32_t first = 32u * j;
    const uint32_t left = end > first ? end - first : 0u;      // bits of dword j before `end`
    return left < 32u ? w | (0xffffffffu >> left) : w;
  }
  __device__ __forceinline__ uint32_t take (uint32_t j)
  {
    if ((j >> 2) != pidx) {
      res = nxt;
      pidx++;
      // (the load must not be scheduled above the copy: its result would then need registers of
      // its own, and the move into nxt -- with a wait for the load -- would come right behind it)
      asm volatile ("" : "+v" (res) : : "memory");
      nxt = fetch_piece (pidx + 1);
    }
    return guarded (BitReader::pick (res, j & 3u), j);
  }
  __device__ __forceinline__ void start (const BitReader & b, uint32_t p, uint32_t e)
  {
    base = b.base;
    last_piece = b.last_piece;
    pos = p;
    end = e;
    widx = p >> 5;
    pidx = widx >> 2;
    res = fetch_piece (pidx);
    nxt = fetch_piece (pidx + 1);
    w0 = guarded (BitReader::pick (res, widx & 3u), widx);
    w1 = take (widx + 1);
  }
  __device__ __forceinline__ uint32_t peek ()
  {
    const uint32_t wi = pos >> 5;
    if (wi != widx) {
      w0 = w1;
      w1 = take (wi + 1);
      widx = wi;
    }
    return (uint32_t) (((((uint64_t) w0) << 32 | w1) << (pos & 31u)) >> 32);
  }
  __device__ __forceinline__ uint32_t bit ()
  {
    const uint32_t v = peek () >> 31;
    pos++;
    return v;
  }
  // any code (BitReader::sint)
  __device__ __forceinline__ int32_t sint ()
  {
    const uint32_t v = peek ();
    const uint32_t stop = v & 0xaaaaaaaau;
    const int k = __clz ((int) stop);
    if (k <= 30) {
      const int c = k >> 1;
      if (c == 0) {
        pos += 1;
        return 0;
      }
      const uint32_t mag = (1u << c) - 1u + even_bits (v >> (32 - k));
      const uint32_t neg = (v >> (30 - k)) & 1u;
      pos += k + 2;
      return neg ? -(int32_t) mag : (int32_t) mag;
    }
    uint32_t count = 0, value = 0;
    while (!bit ()) {
      count++;
      value = (value << 1) | bit ();
    }
    value += (count < 32u ? 1u << count : 0u) - 1u;
    if (value && bit ())
      value = 0u - value;
    return (int32_t) value;
  }
  // bits 0, 2, 4 ... 30 of t packed into bits 0 ... 15
  static __device__ __forceinline__ uint32_t even_bits (uint32_t t)
  {
    t &= 0x55555555u;
    t = (t | (t >> 1)) & 0x33333333u;
    t = (t | (t >> 2)) & 0x0f0f0f0fu;
    t = (t | (t >> 4)) & 0x00ff00ffu;
    return (t | (t >> 8)) & 0x0000ffffu;
  }
};

__device__ __forceinline__ void
wave_sync ()
{
  __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier ();
  __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront");
}

// one staging turn of a lane: values 0 .. n - 1 of its string, the non-zero ones into mine[]
template < int ARITH >
__device__ __forceinline__ void
run_turn (RunReader & r, int32_t * mine, int n, uint32_t qf, uint32_t qo)
{
  int i = 0;
  while (i < n) {
    const uint32_t v = r.peek ();
    const uint32_t ones = (uint32_t) __clz ((int) ~v);          // 32 when the window is all ones
    const uint32_t z = min (ones, (uint32_t) (n - i));
    i += (int) z;
    r.pos += z;
    if (i < n) {                // (then z == ones: the bit behind the zeros is a 0, or z == 32)
      const uint32_t u = v << (z & 31u);
      const uint32_t k = (uint32_t) __clz ((int) (u & 0xaaaaaaaau));    // 2 * data bits, >= 2 when z < 32
      int32_t d;
      if (__builtin_expect (z + k <= 30u, 1)) { // the code ends inside this window
        const uint32_t mag = (1u << (k >> 1)) - 1u + RunReader::even_bits (u >> (32u - k));      // never 0
        const bool neg = (u >> (30u - k)) & 1u;
        r.pos += k + 2u;
        if constexpr (ARITH == SCHRO_HIP_LOWDELAY_FAST16) {
          d = dequant_s16 (neg ? -(int32_t) mag : (int32_t) mag, qf, qo);
        } else {
          // dequant_int of a magnitude below 2^16 (factor < 2^18): a 24-bit multiply, modulo 2^32
          uint32_t a_u;
          asm ("v_mad_u32_u24 %0, %1, %2, %3" : "=v" (a_u) : "v" (mag), "v" (qf), "v" (qo + 2u));
          const int32_t a = (int32_t) a_u >> 2;
          d = neg ? -a : a;
        }
      } else {
        const int32_t q = r.sint ();
        d = ARITH == SCHRO_HIP_LOWDELAY_FAST16 ? dequant_s16 (q, qf, qo) : dequant_int (q, qf, qo);
      }
      mine[i] = d;
      i++;
    }
  }
}

// one component string (K = 0: luma; K = 1: U and V interleaved) of the wave's slices
template < typename T, int ARITH, int K >
__device__ __forceinline__ void
run_string (RunReader & r, const SliceJob & job, const SliceParams & P, int32_t * stage, uint64_t (*rowbase)[64],
    const uint32_t (*quant)[64], int lane, int sx, int sy, int nvalid, bool active, int base_index)
{
  const int kRunCap = P.run_cap, kRunPitch = kRunCap + 1;
  constexpr int E = 16 / (int) sizeof (T);
  constexpr int C = K + 1;
  const int iwt_w = K ? P.iwt_cw : P.iwt_lw, iwt_h = K ? P.iwt_ch : P.iwt_lh;
  uint8_t *const plane[2] = { (uint8_t *) (K ? job.comp[1] : job.comp[0]), (uint8_t *) job.comp[2] };
  const int plane_stride[2] = { K ? job.stride[1] : job.stride[0], job.stride[2] };
  const int nsub = 1 + 3 * P.depth;
  int32_t *mine = stage + lane * kRunPitch;
#pragma unroll 1
  for (int i = 0; i < nsub; i++) {
    const int qi = min (max (base_index - P.quant_matrix[i], 0), 60);
    const uint32_t qf = quant[0][qi], qo = quant[1][qi];        // (from LDS: a memory round trip per sub-band otherwise)
    const int position = subband_position (i);
    const int shift = P.depth - (position >> 2);
    const int w = iwt_w >> shift, h = iwt_h >> shift;
    const int ebw = w / P.nh, ebh = h / P.nv;           // (the host checked: no remainders)
    size_t pitch[2];
    uint8_t *dst[2];
#pragma unroll
    for (int c = 0; c < C; c++) {
      pitch[c] = (size_t) plane_stride[c] << shift;
      dst[c] = plane[c] + ((position & 2) ? pitch[c] >> 1 : 0) + ((position & 1) ? (size_t) w * sizeof (T) : 0)
          + (size_t) (ebh * sy) * pitch[c] + (size_t) (ebw * sx) * sizeof (T);
    }
    const bool rows_out = ebw % E == 0;
    if (rows_out) {
#pragma unroll
      for (int c = 0; c < C; c++)
        rowbase[c][lane] = (uint64_t) (uintptr_t) dst[c];
    }
    const int rows_turn = min (ebh, kRunCap / (C * ebw));
    const int cpl = ebw / E;
    const uint32_t m_cpl = div_magic (max (cpl, 1));
    const int nchunk = nvalid * cpl;
#pragma unroll 1
    for (int y0 = 0; y0 < ebh; y0 += rows_turn) {
      const int rows = min (rows_turn, ebh - y0);
      run_turn < ARITH > (r, mine, C * rows * ebw, qf, qo);
      wave_sync ();
      if (rows_out) {
#pragma unroll 1
        for (int y = 0; y < rows; y++) {
#pragma unroll
          for (int c = 0; c < C; c++) {
#pragma unroll 1
            for (int ch = lane; ch < nchunk; ch += 64) {
              const int src = mdiv (ch, cpl, m_cpl), xc = ch - src * cpl;
              int32_t *from = stage + src * kRunPitch + C * (y * ebw + xc * E) + c;
              uint32_t e[E];
#pragma unroll
              for (int j = 0; j < E; j++)
                e[j] = (uint32_t) from[C * j];
#pragma unroll
              for (int j = 0; j < E; j++)
                from[C * j] = 0;
              u32x4 o;
              if constexpr (E == 4)
                o = u32x4 { e[0], e[1], e[2], e[3] };
              else
                o = u32x4 { (e[0] & 0xffffu) | (e[1] << 16), (e[2] & 0xffffu) | (e[3] << 16),
                  (e[4] & 0xffffu) | (e[5] << 16), (e[6] & 0xffffu) | (e[7] << 16) };
              uint8_t *to = (uint8_t *) (uintptr_t) rowbase[c][src] + (size_t) (y0 + y) * pitch[c] + (size_t) xc * 16;
              __builtin_nontemporal_store (o, (SCHRO_GLOBAL u32x4 *) to);
            }
          }
        }
      } else {
        // a rectangle narrower than 16 bytes: each lane its own values
#pragma unroll 1
        for (int y = 0; y < rows; y++) {
#pragma unroll 1
          for (int x = 0; x < ebw; x++) {
#pragma unroll
            for (int c = 0; c < C; c++) {
              int32_t *from = mine + C * (y * ebw + x) + c;
              const int32_t d = *from;
              *from = 0;
              if (active)
                gstore < T > ((T *) (dst[c] + (size_t) (y0 + y) * pitch[c]) + x, (T) d);
            }
          }
        }
      }
      wave_sync ();             // the next turn writes the staging
    }
  }
}

constexpr int kRunWaves = 6;
template < typename T, int ARITH >
__global__ __launch_bounds__ (64) __attribute__ ((amdgpu_waves_per_eu (kRunWaves, kRunWaves)))
void slice_run_kernel (const SliceJob * __restrict__ jobs, const SliceParams P)
{
  extern __shared__ int32_t stage[];    // 64 * (P.run_cap + 1) words
  __shared__ uint64_t rowbase[2][64];
  __shared__ uint32_t quant[2][64];     // kQuant
  const SliceJob job = jobs[blockIdx.y];
  const int lane = (int) threadIdx.x;
  const int nslices = P.nh * P.nv;
  const int s0 = blockIdx.x * 64, s = s0 + lane;
  const bool active = s < nslices;
  const int nvalid = min (64, nslices - s0);
  const int sc = active ? s : nslices - 1;
  const int sy = sc / P.nh, sx = sc - sy * P.nh;
  const uint32_t wraps = (uint32_t) (((uint64_t) sc * (uint32_t) P.remainder) / (uint32_t) P.denom);
  const uint32_t wraps1 = (uint32_t) (((uint64_t) (sc + 1) * (uint32_t) P.remainder) / (uint32_t) P.denom);
  const uint32_t offset = (uint32_t) sc * (uint32_t) P.n_bytes + wraps;
  const uint32_t slice_bytes = (uint32_t) P.n_bytes + (wraps1 - wraps);

  for (int j = lane; j < 64 * (P.run_cap + 1); j += 64)
    stage[j] = 0;
  quant[0][lane] = kQuant.factor[min (lane, 60)];
  quant[1][lane] = kQuant.offset[min (lane, 60)];

  // the slice header, with slice_kernel's reader
  const uintptr_t addr = (uintptr_t) job.data;
  BitReader hb;
  hb.base = (const uint32_t *) (addr & ~(uintptr_t) 15);
  const uint32_t lead = 8u * (uint32_t) (addr & 15);
  const uint32_t buffer_end = lead + 8u * job.data_bytes;
  hb.last_piece = (buffer_end - 1u) >> 7;
  const uint32_t slice_end = active ? lead + 8u * (offset + slice_bytes) : 0u;
  hb.end = slice_end;
  hb.start (lead + 8u * offset);
  const int base_index = (int) hb.bits (7);
  const uint32_t field = 8u * (ARITH == SCHRO_HIP_LOWDELAY_FAST16 ? (uint32_t) P.n_bytes : slice_bytes);
  const int length_bits = field ? 32 - __clz ((int) field) : 0;
  const uint32_t slice_y_length = hb.bits (length_bits);
  const uint32_t y_pos = hb.pos;
  const uint32_t y_end = active ? min (y_pos + slice_y_length, buffer_end) : 0u;
  const uint32_t uv_pos = y_pos + slice_y_length;
  wave_sync ();

  const int k_first = gridDim.z == 2 ? (int) blockIdx.z : 0, k_last = gridDim.z == 2 ? (int) blockIdx.z : 1;
  RunReader r;
  if (k_first == 0) {
    r.start (hb, y_pos, y_end);
    run_string < T, ARITH, 0 > (r, job, P, stage, rowbase, quant, lane, sx, sy, nvalid, active, base_index);
  }
  if (k_last == 1) {
    r.start (hb, uv_pos, slice_end);
    run_string < T, ARITH, 1 > (r, job, P, stage, rowbase, quant, lane, sx, sy, nvalid, active, base_index);
  }
}

// ---- DC prediction ---------------------------------------------------------------------
// x[j][i] += pred (x[j][i-1], x[j-1][i], x[j-1][i-1]): serial along rows AND columns, only
// the anti-diagonals are independent.  One workgroup per band; thread r owns row band0 + r
// and is one column behind thread r - 1, so what it needs from the row above was made in
// the previous step (handed over through LDS) and the step before (kept in a register).
constexpr int kDcRows = 576;        // rows of a band = threads of the workgroup (LDS: 128 bytes of window per row)

template < typename T > __device__ __forceinline__ int32_t dc_mean3 (int32_t a);
template <> __device__ __forceinline__ int32_t dc_mean3 < int16_t > (int32_t a)
{
  return (a * 21845 + 10922) >> 16;     // schro_divide3, schroutils.h:64
}
template <> __device__ __forceinline__ int32_t dc_mean3 < int32_t > (int32_t a)
{
  // schro_divide (a, 3), schroutils.h:63: floor
  const int32_t n = a < 0 ? (int32_t) ((uint32_t) a - 2u) : a;
  return n / 3;
}

// sample e of a 16-byte piece / the piece with sample e replaced (e is a compile-time index)
template < typename T, int e > __device__ __forceinline__ int32_t
piece_get (const u32x4 & v)
{
  if constexpr (sizeof (T) == 4)
    return (int32_t) v[e];
  else
    return (int16_t) (v[e >> 1] >> (16 * (e & 1)));
}

template < typename T, int e > __device__ __forceinline__ void
piece_set (u32x4 & v, int32_t s)
{
  if constexpr (sizeof (T) == 4)
    v[e] = (uint32_t) s;
  else if constexpr ((e & 1) == 0)
    v[e >> 1] = (uint32_t) s & 0xffffu;
  else
    v[e >> 1] |= (uint32_t) s << 16;
}

struct DcRow {
  int32_t left, upleft;
};

// one 16-byte piece of a row: E samples, serially
template < typename T, int e = 0 >
__device__ __forceinline__ void
dc_piece (const u32x4 & in, const u32x4 & up, u32x4 & out, DcRow & st, bool first_row, bool first_piece)
{
  constexpr int E = 16 / (int) sizeof (T);
  if constexpr (e < E) {
    const int32_t q = piece_get < T, e > (in);
    int32_t v;
    if (first_row) {
      v = (e > 0 || !first_piece) ? (int32_t) (T) ((uint32_t) q + (uint32_t) st.left) : q;
    } else {
      const int32_t u = piece_get < T, e > (up);
      const int32_t pred = (e == 0 && first_piece) ? u
          : dc_mean3 < T > ((int32_t) ((uint32_t) st.left + (uint32_t) u + (uint32_t) st.upleft + 1u));
      v = (int32_t) (T) ((uint32_t) q + (uint32_t) pred);
      st.upleft = u;
    }
    st.left = v;
    piece_set < T, e > (out, v);
    dc_piece < T, e + 1 > (in, up, out, st, first_row, first_piece);
  }
}

template < typename T, int e = 0 >
__device__ __forceinline__ void
store_some (T * to, const u32x4 & v, int n)
{
  constexpr int E = 16 / (int) sizeof (T);
  if constexpr (e < E) {
    if (e < n)
      gstore < T > (to + e, (T) piece_get < T, e > (v));
    store_some < T, e + 1 > (to, v, n);
  }
}

// workgroup barrier that orders LDS only: __syncthreads () would also wait for the row
// pieces in flight (it fences global memory too)
__device__ __forceinline__ void
lds_barrier ()
{
  __builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier ();
  __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Thread r owns row band0 + r and works in 16-byte pieces (E samples), one piece per step,
// one piece behind thread r - 1: the piece of the row above was finished in the previous
// step and comes through LDS (xv).
//
// Memory side.  The rows of an LL band lie 2^depth frame rows apart, so a wave whose lanes
// each fetch or store the piece of their own row touches 64 cache lines per instruction;
// with one such load and store per step the texture path, at about a line per clock, was
// the whole step (0.85 us).  Rows are therefore moved in groups of kDcGroup pieces -- 128
// contiguous bytes per row, 8 lanes per row, 8 rows per instruction -- through an LDS window
// slot[row][k]: at the top of a group every thread issues the loads of its share of the
// NEXT group's window into registers (two thirds of a microsecond of L2 latency that the 8
// steps of the group cover), the steps work in place on the window, and the end of the
// group flushes the window to memory and refills it from the registers.  The window of
// band row ri for the group that starts at step S0 holds pieces S0 - ri ... + 7 (the pieces
// thread ri works on in those steps); staged row 0 is the row above the band (pieces S0 ...,
// read by thread 0 of a later band, never flushed).
constexpr int kDcGroup = 8;
constexpr int kDcShare = kDcGroup + 1;  // window slots per thread: (rows + 1) * kDcGroup / rows, rounded up
constexpr int kDcPitch = kDcGroup + 1;  // slots per window row in LDS: 144 bytes, so that the lanes' own slots spread over the banks

template < typename T >
struct DcBand {
  static constexpr int E = 16 / (int) sizeof (T);
  int w, rows, npieces, steps;
  int r;
  // this thread's share of the window: slot q = r + R * m lives in staged row q / kDcGroup
  // (0: the row above the band) at position q % kDcGroup -- fixed for the band, worked out once
  T *row[kDcShare];
  int pbase[kDcShare];          // piece held by the slot in the group that starts at step 0
  int lslot[kDcShare];          // slot index, or -1: beyond the window / the row above (never flushed)
  int kpos[kDcShare];

  __device__ __forceinline__ void setup (const DcJob & job, int band0, int R)
  {
#pragma unroll
    for (int m = 0; m < kDcShare; m++) {
      const int q = r + R * m, qc = min (q, (rows + 1) * kDcGroup - 1);
      const int i = qc / kDcGroup, k = qc - i * kDcGroup, ri = i - 1;
      row[m] = (T *) ((uint8_t *) job.data + (size_t) min (max (band0 + ri, 0), job.h - 1) * job.stride);
      pbase[m] = (i == 0 ? 0 : -ri) + k;
      kpos[m] = k;
      const int at = i * kDcPitch + k;
      lslot[m] = q < (rows + 1) * kDcGroup ? (i == 0 ? -2 - at : at) : -1;  // -2 - at: fill only
    }
  }
  __device__ __forceinline__ void load_window (int S0, u32x4 * regs) const
  {
#pragma unroll
    for (int m = 0; m < kDcShare; m++)
      regs[m] = gload < u32x4 > (row[m] + (size_t) min (max (pbase[m] + S0, 0), npieces - 1) * E);
  }
  // window of the group that started at S0 out to memory, the next group's (regs) in
  __device__ __forceinline__ void turn_window (u32x4 * slot, int S0, const u32x4 * regs, bool flush) const
  {
#pragma unroll
    for (int m = 0; m < kDcShare; m++) {
      const int q = lslot[m];
      if (q == -1)
        continue;
      const int p = pbase[m] + S0;
      // (the slots that a step of this group worked on)
      if (flush && q >= 0 && p >= 0 && p < npieces && S0 + kpos[m] < steps) {
        const u32x4 v = slot[q];
        T *to = row[m] + (size_t) p * E;
        const int n = w - p * E;
        if (n >= E)
          gstore < u32x4 > (to, v);
        else
          store_some < T > (to, v, n);  // the band's row ends inside this piece
      }
    }
    lds_barrier ();             // every slot read before it is refilled
#pragma unroll
    for (int m = 0; m < kDcShare; m++) {
      const int q = lslot[m];
      if (q != -1)
        slot[q >= 0 ? q : -2 - q] = regs[m];
    }
    lds_barrier ();
  }
};

template < typename T, int k = 0 >
__device__ __forceinline__ void
dc_group_steps (const DcBand < T > &b, int j, int S0, u32x4 * slot, u32x4 (*xv)[kDcRows], DcRow & st)
{
  if constexpr (k < kDcGroup) {
    const int S = S0 + k;
    if (S < b.steps) {          // uniform
      const int p = S - b.r;
      if (b.r < b.rows && p >= 0 && p < b.npieces) {
        u32x4 *mine = slot + (b.r + 1) * kDcPitch + k;
        u32x4 up = { 0, 0, 0, 0 }, out = { 0, 0, 0, 0 };
        if (j > 0)
          up = b.r == 0 ? slot[k] : xv[(S + 1) & 1][b.r - 1];
        dc_piece < T > (*mine, up, out, st, j == 0, p == 0);
        *mine = out;
        xv[S & 1][b.r] = out;
      }
      lds_barrier ();
    }
    dc_group_steps < T, k + 1 > (b, j, S0, slot, xv, st);
  }
}

template < typename T >
__global__ __launch_bounds__ (kDcRows)
void dc_predict_kernel (const DcJob * __restrict__ jobs)
{
  __shared__ u32x4 slot[(kDcRows + 1) * kDcPitch];
  __shared__ u32x4 xv[2][kDcRows];
  constexpr int E = 16 / (int) sizeof (T);
  const DcJob job = jobs[blockIdx.x];
  const int r = threadIdx.x, R = blockDim.x;
  const int w = job.w, h = job.h;
  if ((((uintptr_t) job.data | (uintptr_t) job.stride) & 15) == 0) {
    // (a stride that is a multiple of 16 also means whole pieces can be read at a row's end)
    for (int band0 = 0; band0 < h; band0 += R) {
      DcBand < T > b;
      b.w = w;
      b.rows = min (R, h - band0);
      b.npieces = (w + E - 1) / E;
      b.steps = b.npieces + b.rows - 1;
      b.r = r;
      b.setup (job, band0, R);
      const int j = band0 + r;
      DcRow st = { 0, 0 };
      u32x4 regs[kDcShare];
      b.load_window (0, regs);
      b.turn_window (slot, 0, regs, false);
      for (int S0 = 0; S0 < b.steps; S0 += kDcGroup) {
        b.load_window (S0 + kDcGroup, regs);
        dc_group_steps < T > (b, j, S0, slot, xv, st);
        b.turn_window (slot, S0, regs, true);
      }
      __syncthreads ();         // the band's rows are in memory before the next band reads its last one
    }
    return;
  }
  // ---- planes that are not 16-byte aligned: sample by sample -------------------------------
  int32_t (*xchg)[kDcRows] = reinterpret_cast < int32_t (*)[kDcRows] > (&xv[0][0]);
  for (int band0 = 0; band0 < h; band0 += R) {
    const int j = band0 + r;
    const bool have_row = j < h;
    T *line = (T *) ((uint8_t *) job.data + (size_t) j * job.stride);
    const T *above = (const T *) ((const uint8_t *) job.data + (size_t) (j - 1) * job.stride);
    int32_t left = 0, upleft = 0;
    const int rows = min (R, h - band0);
    const int steps = w + rows - 1;
    for (int s = 0; s < steps; s++) {
      const int x = s - r;
      if (have_row && x >= 0 && x < w) {
        const int32_t q = gload < T > (line + x);
        int32_t v;
        if (j == 0) {
          v = x > 0 ? (int32_t) (T) ((uint32_t) q + (uint32_t) left) : q;
        } else {
          // the row above: finished by the previous band (global) or one step ahead (LDS)
          const int32_t up = r == 0 ? (int32_t) gload < T > (above + x) : xchg[(s + 1) & 1][r - 1];
          const int32_t pred = x == 0 ? up
              : dc_mean3 < T > ((int32_t) ((uint32_t) left + (uint32_t) up + (uint32_t) upleft + 1u));
          v = (int32_t) (T) ((uint32_t) q + (uint32_t) pred);
          upleft = up;
        }
        gstore < T > (line + x, (T) v);
        left = v;
        xchg[s & 1][r] = v;
      }
      __syncthreads ();
    }
    __syncthreads ();
  }
}

// slice_run_kernel's conditions: every sub-band divides evenly into the slices and a row of a
// slice rectangle (U and V together) fits a staging turn; the words per lane it then takes, or 0
static int
slice_run_cap (const SliceParams & P)
{
  int widest = 0;
  for (int k = 0; k < 2; k++) {
    for (int level = 0; level <= P.depth; level++) {
      const int shift = level == 0 ? P.depth : P.depth - level + 1;     // LL and level 1 have the same size
      const int w = (k ? P.iwt_cw : P.iwt_lw) >> shift, h = (k ? P.iwt_ch : P.iwt_lh) >> shift;
      if (w <= 0 || h <= 0 || w % P.nh || h % P.nv)
        return 0;
      widest = std::max (widest, (k + 1) * (w / P.nh));
    }
  }
  for (int cap = kRunCapMin; cap <= kRunCapMax; cap *= 2)
    if (widest <= cap)
      return cap;
  return 0;
}

int
launch_slices (hipStream_t stream, const SliceJob * d_jobs, int njobs, const SliceParams & P0, int bpp, int arith,
    bool aligned16)
{
  SliceParams P = P0;
  // (SCHRO_HIP_SLICE_SPLIT=0: one workgroup decodes both strings of its slices;
  //  SCHRO_HIP_SLICE_RUNS=0: slice_kernel -- a step per value -- for every geometry)
  static const bool split = !SCHRO_ENV ("SCHRO_HIP_SLICE_SPLIT") || atoi (SCHRO_ENV ("SCHRO_HIP_SLICE_SPLIT")) != 0;
  const char *env = SCHRO_ENV ("SCHRO_HIP_SLICE_RUNS");
  P.run_cap = aligned16 && !(env && atoi (env) == 0) ? slice_run_cap (P) : 0;
  const bool runs = P.run_cap != 0;
  const size_t lds = 64 * (size_t) (P.run_cap + 1) * sizeof (int32_t);
  const dim3 grid ((unsigned) ((P.nh * P.nv + 63) / 64), (unsigned) njobs, split ? 2u : 1u);
  if (runs) {
    if (bpp == 4)
      SCHRO_LAUNCH ((slice_run_kernel < int32_t, SCHRO_HIP_LOWDELAY_S32 >), grid, dim3 (64), lds, stream, d_jobs, P);
    else if (arith == SCHRO_HIP_LOWDELAY_FAST16)
      SCHRO_LAUNCH ((slice_run_kernel < int16_t, SCHRO_HIP_LOWDELAY_FAST16 >), grid, dim3 (64), lds, stream, d_jobs, P);
    else
      SCHRO_LAUNCH ((slice_run_kernel < int16_t, SCHRO_HIP_LOWDELAY_SLOW16 >), grid, dim3 (64), lds, stream, d_jobs, P);
  } else if (bpp == 4)
    SCHRO_LAUNCH ((slice_kernel < int32_t, SCHRO_HIP_LOWDELAY_S32 >), grid, dim3 (64), 0, stream, d_jobs, P);
  else if (arith == SCHRO_HIP_LOWDELAY_FAST16)
    SCHRO_LAUNCH ((slice_kernel < int16_t, SCHRO_HIP_LOWDELAY_FAST16 >), grid, dim3 (64), 0, stream, d_jobs, P);
  else
    SCHRO_LAUNCH ((slice_kernel < int16_t, SCHRO_HIP_LOWDELAY_SLOW16 >), grid, dim3 (64), 0, stream, d_jobs, P);
  const hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "slice kernel launch: %s", hipGetErrorString (e));
  return 0;
}

// ---- r03: dc_skew_kernel -- the same prediction, one sample per step, strips on separate CUs ----
// dc_predict_kernel steps by 16-byte pieces with a workgroup barrier per step: 0.5 us per step, 780
// steps for the 960x540 luma band of config 5.  Here a strip of 64 rows is a workgroup of two waves:
//  * the COMPUTE wave: lane r = row r of the strip, one SAMPLE behind lane r - 1, so what a lane needs
//    from the row above is the neighbour lane's result of the step before (one DPP wave shift -- no
//    LDS exchange, no barrier) and its own `up` of the step before.  It touches LDS only, a block of
//    16 samples at a time: its rows come out of a ring of 16 blocks per row, its results go into a
//    ring of 8.  A lone wave issues an instruction every ~6 cycles whatever it is, so a step is
//    priced in instructions: 8 for s32 (DPP, two adds, the floor division by 3 in five, the sample);
//  * the MOVER wave keeps the rings going: 4 blocks per row on their way from memory at any time
//    (the compute wave takes a block every ~0.8 us, less than a memory round trip), finished blocks
//    out to the band, the strip's last row to the strip below, the last row of the strip above in.
//    The two meet in three LDS counters (blocks ready / blocks done / blocks stored); a wave's LDS
//    operations execute in order, so a counter written behind the data is seen behind the data.
// The strips of a band are workgroups on different CUs.  Strip k + 1 takes the last row of strip k
// from a hand-over buffer in which every sample carries the launch's epoch (one 64-bit relaxed
// device-scope store per sample: a sample is valid when its tag is, no fence), and runs 64 samples
// + the hand-over behind it.  Strips are dispatched in order, so a strip that waits always waits
// for one that is already running.  w + h + (hand-overs) steps of ~10 instructions instead of
// w / E + h steps of a workgroup barrier each.
constexpr int kSkewBlock = 16;  // samples per block
constexpr int kSkewIn = 16;     // ring of the rows: blocks per row (+ 1: block 0 again behind the last, for reads across the end)
constexpr int kSkewOut = 8;     // ring of the results (+ 1: samples written across the end)
constexpr int kSkewDepth = 4;   // blocks per row on their way from memory
constexpr int kSkewPatience = 1 << 20;  // times a mover asks for the strip above's samples (~2 us each) before it gives up

template < typename T >
struct SkewRings {
  static constexpr int G = kSkewBlock * (int) sizeof (T) / 16;  // 16-byte pieces per block
  u32x4 in[64][(kSkewIn + 1) * G];
  u32x4 out[64][(kSkewOut + 1) * G];
  int32_t above[kSkewIn * kSkewBlock];  // the row above the strip (lane 0's `up`)
  int ready;                    // blocks of every row (and of the row above) in the ring
  int done;                     // blocks the compute wave has finished
  int stored;                   // blocks of every row out of the ring
  int abort;                    // the mover gave up waiting for the strip above: both waves leave
};

__device__ __forceinline__ int
lds_peek (const int *p)
{
  const int v = __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile ("" : : : "memory");
  return v;
}

__device__ __forceinline__ void
lds_post (int *p, int v)
{
  asm volatile ("" : : : "memory");     // (behind the data in program order; the LDS keeps a wave's order)
  __hip_atomic_store (p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// one sample of every lane.  EDGE: lanes may be before their row's first or behind its last sample.
template < typename T, bool FIRST_STRIP, bool EDGE >
__device__ __forceinline__ int32_t
dc_skew_step (int32_t q, int32_t from_above, int32_t & left, int32_t & upleft, int x, int w_lane, bool first_row)
{
  // the neighbour lane's result of the step before = this column of the row above (lane 0: handed over)
  const int32_t up = __builtin_amdgcn_update_dpp (from_above, left, 0x138 /* wave_shr:1 */ , 0xf, 0xf, false);
  const int32_t a = (int32_t) ((uint32_t) up + (uint32_t) upleft + 1u + (uint32_t) left);
  int32_t mean;
  if constexpr (sizeof (T) == 4) {
    // dc_mean3 < int32_t > without a compare: n = a < 0 ? a - 2 : a (wrapping), n / 3 truncated
    const int32_t n = (int32_t) ((uint32_t) a + ((uint32_t) (a >> 31) << 1));
    mean = __mulhi (n, 0x55555556) + (int32_t) ((uint32_t) n >> 31);
  } else {
    mean = dc_mean3 < T > (a);
  }
  int32_t pred = mean;
  if (EDGE)
    pred = x == 0 ? up : pred;
  if (FIRST_STRIP)
    pred = first_row ? left : pred;
  const int32_t v = (int32_t) (T) ((uint32_t) q + (uint32_t) pred);
  upleft = up;
  if (EDGE)
    left = (unsigned) x < (unsigned) w_lane ? v : left;
  else
    left = v;
  return v;
}

template < typename T, bool FIRST_STRIP >
__device__ __forceinline__ void
dc_skew_compute (SkewRings < T > &sh, const DcJob & job, int strip)
{
  constexpr int B = kSkewBlock, SZ = (int) sizeof (T);
  const int lane = (int) threadIdx.x;
  const int w = job.w, nblocks = (w + B - 1) / B;
  const int j = strip * 64 + lane;
  const bool first_row = j == 0;
  const int w_lane = j < job.h ? w : 0;
  const uint8_t *const my_in = reinterpret_cast < const uint8_t * >(&sh.in[lane][0]);
  uint8_t *const my_out = reinterpret_cast < uint8_t * >(&sh.out[lane][0]);
  int32_t left = 0, upleft = 0;
  int32_t q[B], a[B];           // this block's samples of the lane's row / of the row above the strip
#pragma unroll
  for (int e = 0; e < B; e++)
    q[e] = a[e] = 0;
  int ready = 0, stored = 0;    // the counters as last seen
  // lane 63 finishes its row 63 steps behind lane 0; the mover stores a block when lane 63 is through
  // with it, which it sees from `done` being 5 blocks further
  const int turns = nblocks + 4;
#pragma unroll 1
  for (int b = -1; b < turns; b++) {
    // ---- the next block's samples out of the rings (lane 0 is in block b + 1 then) ------------------
    const int need = min (b + 2, nblocks);
    while (ready < need) {
      ready = lds_peek (&sh.ready);
      if (ready < need) {
        if (lds_peek (&sh.abort))
          return;
        __builtin_amdgcn_s_sleep (1);
      }
    }
    int32_t qn[B], an[B];
    const int xn = (b + 1) * B - lane;          // this lane's sample at the next block's first step
    const uint8_t *from = my_in + ((xn * SZ) & (kSkewIn * B * SZ - 1)); // (B samples from here: up to a block across the ring's end)
#pragma unroll
    for (int e = 0; e < B; e++) {
      qn[e] = (int32_t) * reinterpret_cast < const T * >(from + e * SZ);
      an[e] = FIRST_STRIP ? 0 : sh.above[((b + 1) & (kSkewIn - 1)) * B + e];
    }
    if (b >= 0) {
      // ---- room for this block's results: the block that had their place is out of the ring ----------
      while (stored < b - (kSkewOut - 1)) {
        stored = lds_peek (&sh.stored);
        if (stored < b - (kSkewOut - 1)) {
          if (lds_peek (&sh.abort))
            return;
          __builtin_amdgcn_s_sleep (1);
        }
      }
      // ---- B steps ----------------------------------------------------------------------------------------
      const int xg = b * B - lane;
      uint8_t *to = my_out + ((xg * SZ) & (kSkewOut * B * SZ - 1));
      if (b * B >= 63 && b * B + B <= w) {      // every lane is inside its row for the whole block
#pragma unroll
        for (int e = 0; e < B; e++)
          *reinterpret_cast < T * >(to + e * SZ) =
              (T) dc_skew_step < T, FIRST_STRIP, false > (q[e], a[e], left, upleft, xg + e, w_lane, first_row);
      } else {
        // (a sample before / behind the row lands in a place that is rewritten before / done with after its store)
#pragma unroll
        for (int e = 0; e < B; e++)
          *reinterpret_cast < T * >(to + e * SZ) =
              (T) dc_skew_step < T, FIRST_STRIP, true > (q[e], a[e], left, upleft, xg + e, w_lane, first_row);
      }
      lds_post (&sh.done, b + 1);
    }
#pragma unroll
    for (int e = 0; e < B; e++) {
      q[e] = qn[e];
      a[e] = an[e];
    }
  }
}

template < typename T, bool FIRST_STRIP >
__device__ __forceinline__ void
dc_skew_move (SkewRings < T > &sh, const DcJob & job, int strip, unsigned long long *edge_out, int edge_pitch,
    uint32_t epoch, uint32_t * gave_up)
{
  constexpr int B = kSkewBlock, E = 16 / (int) sizeof (T), G = B / E, D = kSkewDepth;
  const int lane = (int) threadIdx.x - 64;
  const int w = job.w, npieces = w / E, nblocks = (w + B - 1) / B;
  const int j = strip * 64 + lane;
  const bool have_row = j < job.h;
  uint8_t *const row = (uint8_t *) job.data + (size_t) min (j, job.h - 1) * job.stride;
  const unsigned long long *const edge_in = edge_out - edge_pitch;      // (strip > 0)
  const bool publish = (strip + 1) * 64 < job.h;
  const unsigned long long tag = (unsigned long long) epoch << 32;
  // A lane writes the first r16 samples of a block a turn before the rest; of the block at the ring's
  // start they went behind the ring's end.
  const int r16 = (B - (lane & (B - 1))) & (B - 1);
  u32x4 across[G];              // all ones in those first r16 samples, piece by piece
#pragma unroll
  for (int g = 0; g < G; g++) {
    uint32_t m[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
      if constexpr (E == 4)
        m[d] = g * E + d < r16 ? 0xffffffffu : 0u;
      else
        m[d] = (g * E + 2 * d < r16 ? 0xffffu : 0u) | (g * E + 2 * d + 1 < r16 ? 0xffff0000u : 0u);
    }
    across[g] = u32x4 { m[0], m[1], m[2], m[3] };
  }
  const T *const last_out = reinterpret_cast < const T * >(&sh.out[63][0]);

  int ns = 0;                   // blocks stored
  // finished blocks (the last row, lane 63, is through with them) out to the band and to the strip below
  auto service =[&]() {
    const int done = lds_peek (&sh.done);
    bool any = false;
    while (ns < nblocks && done >= ns + 5) {
      const int slot = ns & (kSkewOut - 1);
#pragma unroll
      for (int g = 0; g < G; g++) {
        u32x4 v = sh.out[lane][slot * G + g];
        if (slot == 0) {
          const u32x4 o = sh.out[lane][kSkewOut * G + g];
          v = (v & ~across[g]) | (o & across[g]);
        }
        if (have_row && ns * G + g < npieces)
          gstore < u32x4 > ((u32x4 *) (row + ((size_t) ns * G + g) * 16), v);
      }
      if (publish && lane < B && ns * B + lane < w) {
        // (row 63 writes its first sample of a block -- r16 = 1 -- across the end when the block is the ring's first)
        const int at = (slot == 0 && lane < 1 ? kSkewOut : slot) * B + lane;
        const uint32_t s = (uint32_t) (int32_t) last_out[at];
        __hip_atomic_store (edge_out + ns * B + lane, tag | s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      ns++;
      any = true;
    }
    if (any)
      lds_post (&sh.stored, ns);
  };

  u32x4 fly[D][G];              // blocks on their way in
  unsigned long long efly[D];   // and the row above's (lanes 0 .. 15: a sample each)
  auto fetch =[&](int i, int blk) {     // i: compile-time slot of the pipeline
#pragma unroll
    for (int g = 0; g < G; g++)
      if (have_row && blk * G + g < npieces)
        fly[i][g] = gload < u32x4 > ((const u32x4 *) (row + ((size_t) blk * G + g) * 16));
    efly[i] = tag;
    if (!FIRST_STRIP && lane < B && blk * B + lane < w)
      efly[i] = __hip_atomic_load (edge_in + blk * B + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
#pragma unroll
  for (int i = 0; i < D; i++) {
#pragma unroll
    for (int g = 0; g < G; g++)
      fly[i][g] = u32x4 { 0, 0, 0, 0 };
    efly[i] = tag;
    if (i < nblocks)
      fetch (i, i);
  }
  int done = 0;
#pragma unroll 1
  for (int b0 = 0; b0 < nblocks; b0 += D) {
#pragma unroll
    for (int i = 0; i < D; i++) {
      const int blk = b0 + i;
      if (blk < nblocks) {      // uniform
        // the place's last block (blk - kSkewIn) is behind lane 63, which is 4 blocks behind lane 0
        while (blk >= done + kSkewIn - 4) {
          service ();
          done = lds_peek (&sh.done);
          if (blk >= done + kSkewIn - 4)
            __builtin_amdgcn_s_sleep (1);
        }
        if (!FIRST_STRIP) {
          // the strip above has got this far?  If not: ask again, for every block that is on its way
          int asked = 0;
          while (__any ((uint32_t) (efly[i] >> 32) != epoch)) {
            // Every wait of this kernel ends here: the strip above runs already (tickets), so the samples
            // come.  Should they not (a strip lost to a fault): after ~2 s of asking both waves leave, the
            // strips below follow one by one, and the host finds the launch's epoch in *gave_up.
            if (++asked > kSkewPatience) {
              if (lane == 0)
                __hip_atomic_store (gave_up, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              lds_post (&sh.abort, 1);
              return;
            }
            service ();
            __builtin_amdgcn_s_sleep (2);
#pragma unroll
            for (int k = 0; k < D; k++) {
              const int bk = k >= i ? b0 + k : b0 + D + k;      // the block slot k is fetching
              if ((uint32_t) (efly[k] >> 32) != epoch && bk < nblocks)
                efly[k] = __hip_atomic_load (edge_in + bk * B + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          if (lane < B)
            sh.above[(blk & (kSkewIn - 1)) * B + lane] = (int32_t) (uint32_t) efly[i];
        }
        const int slot = blk & (kSkewIn - 1);
#pragma unroll
        for (int g = 0; g < G; g++) {
          sh.in[lane][slot * G + g] = fly[i][g];
          if (slot == 0)
            sh.in[lane][kSkewIn * G + g] = fly[i][g];
        }
        lds_post (&sh.ready, blk + 1);
        if (blk + D < nblocks)
          fetch (i, blk + D);
        service ();
      }
    }
  }
  while (ns < nblocks) {
    service ();
    __builtin_amdgcn_s_sleep (1);
  }
}

// Which strip a workgroup works on is decided when it starts to run (a ticket from a counter in the
// hand-over buffer), not by its index: a strip then only ever waits for strips that are running
// already, whatever the dispatcher's order or however many launches share the CUs.  The last
// workgroup to finish puts the two counters back to 0 for the queue's next launch.
template < typename T >
__global__ __launch_bounds__ (128)
void dc_skew_kernel (const DcJob * __restrict__ jobs, unsigned long long *edge, int edge_pitch, uint32_t epoch, int strips,
    uint32_t * gave_up)
{
  __shared__ SkewRings < T > sh;
  __shared__ int s_ticket;
  unsigned int *const ctrl = reinterpret_cast < unsigned int *>(edge);  // [0] tickets, [1] workgroups finished
  edge += 8;
  if (threadIdx.x == 0) {
    s_ticket = (int) __hip_atomic_fetch_add (&ctrl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh.ready = sh.done = sh.stored = sh.abort = 0;
  }
  __syncthreads ();
  const int ticket = s_ticket, nj = ticket / strips, strip = ticket - nj * strips;
  const DcJob job = jobs[nj];
  if (strip * 64 < job.h) {
    if (threadIdx.x < 64) {
      if (strip == 0)
        dc_skew_compute < T, true > (sh, job, strip);
      else
        dc_skew_compute < T, false > (sh, job, strip);
    } else {
      if (strip == 0)
        dc_skew_move < T, true > (sh, job, strip, edge + ((size_t) nj * strips + strip) * edge_pitch, edge_pitch, epoch, gave_up);
      else
        dc_skew_move < T, false > (sh, job, strip, edge + ((size_t) nj * strips + strip) * edge_pitch, edge_pitch, epoch, gave_up);
    }
  }
  __syncthreads ();
  if (threadIdx.x == 0) {
    const unsigned int total = gridDim.x;
    if (__hip_atomic_fetch_add (&ctrl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1u) {
      __hip_atomic_store (&ctrl[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store (&ctrl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// dc_skew_kernel's conditions: rows of whole 16-byte pieces, 16-byte aligned
bool
dc_skew_ok (const DcJob * jobs, int njobs, int bpp)
{
  const char *env = SCHRO_ENV ("SCHRO_HIP_DC_SKEW");
  if (env && atoi (env) == 0)
    return false;
  for (int p = 0; p < njobs; p++)
    if ((((uintptr_t) jobs[p].data | (uintptr_t) jobs[p].stride) & 15) || (jobs[p].w * bpp) % 16)
      return false;
  return true;
}

int
launch_dc_predict (hipStream_t stream, const DcJob * d_jobs, int njobs, int max_rows, int bpp,
    unsigned long long *edge, int edge_pitch, uint32_t epoch, uint32_t * gave_up)
{
  if (edge) {
    // a workgroup (compute wave + mover wave) per strip of 64 rows of every band
    const int strips = (max_rows + 63) / 64;
    const dim3 grid ((unsigned) (strips * njobs));
    if (bpp == 4)
      SCHRO_LAUNCH ((dc_skew_kernel < int32_t >), grid, dim3 (128), 0, stream, d_jobs, edge, edge_pitch, epoch, strips, gave_up);
    else
      SCHRO_LAUNCH ((dc_skew_kernel < int16_t >), grid, dim3 (128), 0, stream, d_jobs, edge, edge_pitch, epoch, strips, gave_up);
  } else {
    // whole waves; a band taller than kDcRows is walked in slabs
    const int threads = std::min (kDcRows, (max_rows + 63) / 64 * 64);
    if (bpp == 4)
      SCHRO_LAUNCH ((dc_predict_kernel < int32_t >), dim3 (njobs), dim3 (threads), 0, stream, d_jobs);
    else
      SCHRO_LAUNCH ((dc_predict_kernel < int16_t >), dim3 (njobs), dim3 (threads), 0, stream, d_jobs);
  }
  const hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "dc_predict kernel launch: %s", hipGetErrorString (e));
  return 0;
}

}                               // namespace schro
