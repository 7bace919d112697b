// dequant.hip -- core-syntax coefficient reconstruction after entropy decoding.
//
// What it computes, per codeblock of a sub-band (schro_decoder_decode_subband,
// schrodecoder.c:3525-3640): a zero codeblock is filled with zeros (schro_decoder_zero_block
// :3311-3322); any other one gets its quantised values dequantised into the coefficient frame,
//   arith 0: v = sign (q) * ((quant_offset + quant_factor * |q| + 2) >> 2) in C int arithmetic
//            -- the arithmetic-coded line decoders (:3072-3079) and orc_dequantise_s32_ip_2d;
//   arith 1: the 16-bit Orc arithmetic of orc_dequantise_s16_2d_8xn / _4xn / _s16_ip_2d
//            (schroorc.orc:1098-1170), the VLC (is_noarith) path on s16 frames (:3406-3441)
// with quant_factor = schro_table_quant[i], quant_offset = schro_table_offset_1_2[i] (intra) or
// _3_8[i] (inter) (:3400-3405).  The serial part -- binary arithmetic / VLC decoding, whose
// contexts depend only on whether neighbours are zero and on their sign, i.e. on the QUANTISED
// values -- stays on the host; the host hands over quantised values, 1 / 2 / 4 bytes each,
// only for the codeblocks that are not zero: most of a coefficient frame never crosses PCIe.
//
// One 256-thread workgroup = a 64 x 64 sample tile of one codeblock, 4 samples x 4 rows per lane
// (r03; a 64 x 16 tile per workgroup made 390 k waves per 8 x 2160p, each a chain of the two job
// probes, the job, the values and the store -- latency-bound at 0.235 ms);
// coefficient rows of a sub-band are contiguous runs inside the interleaved frame rows, so
// stores are coalesced.  Bound: HBM write (bpp bytes per sample) + the values read.

#include "schro_hip_internal.h"

namespace schro {
namespace {

constexpr int kDqThreads = 256, kDqTW = 64, kDqRows = 4, kDqTH = 16 * kDqRows;

struct QuantTables3 {
  uint32_t factor[61], off12[61], off38[61];
};
constexpr QuantTables3
make_quant_tables3 ()
{
  QuantTables3 t = { };
  for (int q = 0; q <= 60; q++) {
    // Dirac specification 13.3.1; all 61 entries of each table are pinned against the
    // reference's numbers by tests (quant_tables.json, arith_lut.json)
    const uint64_t base = (uint64_t) 1 << (q / 4);
    const uint64_t f = (q & 3) == 0 ? 4 * base : (q & 3) == 1 ? (503829 * base + 52958) / 105917
        : (q & 3) == 2 ? (665857 * base + 58854) / 117708 : (440253 * base + 32722) / 65444;
    t.factor[q] = (uint32_t) f;
    t.off12[q] = q == 0 ? 1u : q == 1 ? 2u : (uint32_t) ((f + 1) / 2);
    t.off38[q] = q == 0 ? 1u : (uint32_t) ((f * 3 + 4) / 8);
  }
  return t;
}
constexpr QuantTables3 kHostQuant = make_quant_tables3 ();
static __device__ __constant__ QuantTables3 kDevQuant = make_quant_tables3 ();     // (plans: the kernel looks the quantiser up)

template < typename T, int ARITH >
__device__ __forceinline__ T
dequant_one (int32_t q, uint32_t factor, uint32_t offset)
{
  if constexpr (ARITH == 1) {
    const int16_t qs = (int16_t) q;
    const int16_t sign = qs > 0 ? 1 : (qs < 0 ? -1 : 0);
    const int16_t mag = (int16_t) (qs < 0 ? -qs : qs);          // absw: -32768 stays -32768
    int16_t t = (int16_t) (mag * (int16_t) factor);
    t = (int16_t) (t + (int16_t) (offset + 2u));
    t = (int16_t) (t >> 2);
    return (T) (int16_t) (t * sign);
  } else {
    if (q == 0)
      return (T) 0;
    const uint32_t mag = q < 0 ? 0u - (uint32_t) q : (uint32_t) q;
    const int32_t d = (int32_t) (mag * factor + offset + 2u) >> 2;
    return (T) (q < 0 ? (int32_t) (0u - (uint32_t) d) : d);
  }
}

// which job owns tile `bid`: three probes of 64 lanes in the dense arrays of first tiles behind the
// jobs (every 4096th job, every 64th of that run, the 64 of that run)
template < typename JOB >
__device__ __forceinline__ int
find_dequant_job (const JOB * jobs, int njobs, int bid)
{
  const int lane = threadIdx.x & 63;
  const int *first = reinterpret_cast < const int *>(jobs + njobs);
  const int n64 = (njobs + 63) / 64;
  int lo = 0;
  if (njobs > 4096) {
    const int idx = lane;
    const bool le = idx * 4096 < njobs && gload < int > (first + njobs + n64 + idx) <= bid;
    lo = (__popcll (__ballot (le)) - 1) * 4096;
  }
  if (njobs > 64) {
    const int idx = (lo >> 6) + lane;
    const bool le = idx < n64 && gload < int > (first + njobs + idx) <= bid;
    lo += (__popcll (__ballot (le)) - 1) * 64;
  }
  const int idx = lo + lane;
  const bool le = idx < njobs && gload < int > (first + idx) <= bid;
  return __builtin_amdgcn_readfirstlane (lo + __popcll (__ballot (le)) - 1);
}

// 4 quantised values of `bytes` bytes each from s (any alignment: one load of 4 * bytes bytes)
template < typename Q >
__device__ __forceinline__ void
load_quads (const void *src, size_t s, int n, int32_t * q)
{
  const Q *p = (const Q *) src + s;
  if (n == 4) {
    if constexpr (sizeof (Q) == 1) {
      // (as a dword at any byte address: the compiler splits a byte-aligned char vector into three loads)
      typedef uint32_t u32_a1 __attribute__ ((aligned (1)));
      const uint32_t v = *(const SCHRO_GLOBAL u32_a1 *) p;
      q[0] = (int32_t) (int8_t) v;
      q[1] = (int32_t) (int8_t) (v >> 8);
      q[2] = (int32_t) (int8_t) (v >> 16);
      q[3] = (int32_t) v >> 24;
    } else {
      typedef Q Q4 __attribute__ ((ext_vector_type (4), aligned (sizeof (Q))));
      const Q4 v = *(const SCHRO_GLOBAL Q4 *) p;
      q[0] = (int32_t) v.x;
      q[1] = (int32_t) v.y;
      q[2] = (int32_t) v.z;
      q[3] = (int32_t) v.w;
    }
  } else {
    for (int e = 0; e < 4; e++)
      q[e] = e < n ? (int32_t) gload < Q > (p + e) : 0;
  }
}

// one tile of one codeblock (job: where it lies and what dequantises it)
template < typename T, int ARITH >
__device__ __forceinline__ void
dequant_tile (const DequantJob & job, int t)
{
  const int ty = t / job.tiles_x, tx = t - ty * job.tiles_x;
  const int x = tx * kDqTW + 4 * (threadIdx.x & 15), y0 = ty * kDqTH + (threadIdx.x >> 4);
  if (x >= job.w)
    return;
  const int n = min (4, job.w - x);
  // a lane's rows lie 16 apart; the values of all of them are asked for before the first is used
  int32_t q[kDqRows][4];
#pragma unroll
  for (int r = 0; r < kDqRows; r++) {
    const int y = y0 + 16 * r;
    q[r][0] = q[r][1] = q[r][2] = q[r][3] = 0;
    if (job.src && y < job.h) {
      const size_t s = (size_t) y * job.w + x;
      if (job.src_bytes == 1)
        load_quads < int8_t > (job.src, s, n, q[r]);
      else if (job.src_bytes == 2)
        load_quads < int16_t > (job.src, s, n, q[r]);
      else
        load_quads < int32_t > (job.src, s, n, q[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < kDqRows; r++) {
    const int y = y0 + 16 * r;
    if (y >= job.h)
      break;
    T v[4] = { 0, 0, 0, 0 };
    if (job.src)
      for (int e = 0; e < 4; e++)
        v[e] = dequant_one < T, ARITH > (q[r][e], job.factor, job.offset);
    T *d = (T *) ((char *) job.dst + (size_t) y * job.dst_stride) + x;
    if (n == 4 && (((uintptr_t) d) & (4 * sizeof (T) - 1)) == 0) {
      if constexpr (sizeof (T) == 2) {
        u32x2 o;
        o.x = (uint32_t) (uint16_t) v[0] | ((uint32_t) (uint16_t) v[1] << 16);
        o.y = (uint32_t) (uint16_t) v[2] | ((uint32_t) (uint16_t) v[3] << 16);
        gstore < u32x2 > (d, o);
      } else {
        u32x4 o;
        o.x = (uint32_t) v[0];
        o.y = (uint32_t) v[1];
        o.z = (uint32_t) v[2];
        o.w = (uint32_t) v[3];
        gstore < u32x4 > (d, o);
      }
    } else {
      for (int e = 0; e < n; e++)
        gstore < T > (d + e, v[e]);
    }
  }
}

template < typename T, int ARITH >
__global__ __launch_bounds__ (kDqThreads)
void dequant_kernel (const DequantJob * __restrict__ jobs, int njobs)
{
  const int bid = blockIdx.x;
  const DequantJob job = jobs[find_dequant_job (jobs, njobs, bid)];
  dequant_tile < T, ARITH > (job, bid - job.tile_base);
}

// r04 -- plans: the geometry comes from the plan's resident table, the per-picture part (does the codeblock
// have values, where, how wide, which quantiser) from the decoder's own records as they were uploaded
template < typename T, int ARITH >
__global__ __launch_bounds__ (kDqThreads)
void dequant_plan_kernel (const DequantGeo * __restrict__ geo, int njobs, const SchroHipCodeblock * __restrict__ recs,
    const DequantPlaneDyn * __restrict__ planes)
{
  const int bid = blockIdx.x;
  const DequantGeo g = geo[find_dequant_job (geo, njobs, bid)];
  const DequantPlaneDyn pd = planes[g.plane];
  const SchroHipCodeblock cb = recs[g.rec];
  DequantJob job;
  job.dst = (char *) pd.dst + g.dst_offset;
  job.src = cb.src_offset < 0 ? nullptr : (const char *) pd.values + cb.src_offset;
  job.dst_stride = g.dst_stride;
  job.w = g.w;
  job.h = g.h;
  job.src_bytes = cb.src_bytes;
  const int q = min ((int) cb.quant_index, 60);
  job.factor = kDevQuant.factor[q];
  job.offset = pd.is_intra ? kDevQuant.off12[q] : kDevQuant.off38[q];
  job.tiles_x = g.tiles_x;
  job.tile_base = g.tile_base;
  dequant_tile < T, ARITH > (job, bid - g.tile_base);
}

}                               // namespace

void
dequant_tile_geometry (int *tw, int *th)
{
  *tw = kDqTW;
  *th = kDqTH;
}

void
dequant_tables (int quant_index, int is_intra, uint32_t * factor, uint32_t * offset)
{
  const int q = quant_index < 0 ? 0 : (quant_index > 60 ? 60 : quant_index);
  *factor = kHostQuant.factor[q];
  *offset = is_intra ? kHostQuant.off12[q] : kHostQuant.off38[q];
}

// A job table from its pinned host mirror to the device by a kernel (16 bytes per lane): a copy
// through the DMA engines queues up behind the coefficient / value uploads of the copy queue --
// 0.8 MB of table waited a millisecond for 54 MB of values.
__global__ __launch_bounds__ (256)
void table_copy_kernel (u32x4 * __restrict__ dst, const u32x4 * __restrict__ src, size_t n16)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if (i < n16)
    dst[i] = src[i];
}

int
launch_table_copy (hipStream_t stream, void *dst, const void *src, size_t bytes)
{
  const size_t n16 = (bytes + 15) / 16;         // (the buffers are allocated in multiples of 16 bytes)
  hipLaunchKernelGGL (table_copy_kernel, dim3 ((unsigned) ((n16 + 255) / 256)), dim3 (256), 0, stream, (u32x4 *) dst,
      (const u32x4 *) src, n16);
  const hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "table copy launch: %s", hipGetErrorString (e));
  return 0;
}

int
launch_dequant_plan (hipStream_t stream, const DequantGeo * d_geo, int njobs, int total_tiles, const SchroHipCodeblock * d_recs,
    const DequantPlaneDyn * d_planes, int bpp, int arith)
{
  if (bpp == 2 && arith == 1)
    SCHRO_LAUNCH ((dequant_plan_kernel < int16_t, 1 >), dim3 (total_tiles), dim3 (kDqThreads), 0, stream, d_geo, njobs, d_recs, d_planes);
  else if (bpp == 2)
    SCHRO_LAUNCH ((dequant_plan_kernel < int16_t, 0 >), dim3 (total_tiles), dim3 (kDqThreads), 0, stream, d_geo, njobs, d_recs, d_planes);
  else
    SCHRO_LAUNCH ((dequant_plan_kernel < int32_t, 0 >), dim3 (total_tiles), dim3 (kDqThreads), 0, stream, d_geo, njobs, d_recs, d_planes);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "dequant (plan) launch: %s", hipGetErrorString (e));
  return 0;
}

int
launch_dequant (hipStream_t stream, const DequantJob * d_jobs, int njobs, int total_tiles, int bpp, int arith)
{
  if (bpp == 2 && arith == 1)
    SCHRO_LAUNCH ((dequant_kernel < int16_t, 1 >), dim3 (total_tiles), dim3 (kDqThreads), 0, stream, d_jobs, njobs);
  else if (bpp == 2)
    SCHRO_LAUNCH ((dequant_kernel < int16_t, 0 >), dim3 (total_tiles), dim3 (kDqThreads), 0, stream, d_jobs, njobs);
  else
    SCHRO_LAUNCH ((dequant_kernel < int32_t, 0 >), dim3 (total_tiles), dim3 (kDqThreads), 0, stream, d_jobs, njobs);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "dequant launch: %s", hipGetErrorString (e));
  return 0;
}

}                               // namespace schro
