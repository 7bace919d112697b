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
// One 256-thread workgroup = a 64 x 16 sample tile of one codeblock, 4 samples per lane;
// coefficient rows of a sub-band are contiguous runs inside the interleaved frame rows, so
// stores are coalesced.  Bound: HBM write (bpp bytes per sample) + the values read.

#include "schro_hip_internal.h"

namespace schro {
namespace {

constexpr int kDqThreads = 256, kDqTW = 64, kDqTH = 16;

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

// which job owns tile `bid`: two probes of 64 lanes (every 64th job, then the 64 of that run)
__device__ __forceinline__ int
find_dequant_job (const DequantJob * jobs, int njobs, int bid)
{
  const int lane = threadIdx.x & 63;
  int lo = 0;
  if (njobs > 64) {
    const int idx = lane * 64;
    const bool le = idx < njobs && gload < int > (&jobs[idx].tile_base) <= bid;
    lo = (__popcll (__ballot (le)) - 1) * 64;
  }
  const int idx = lo + lane;
  const bool le = idx < njobs && gload < int > (&jobs[idx].tile_base) <= bid;
  return __builtin_amdgcn_readfirstlane (lo + __popcll (__ballot (le)) - 1);
}

template < typename T, int ARITH >
__global__ __launch_bounds__ (kDqThreads)
void dequant_kernel (const DequantJob * __restrict__ jobs, int njobs)
{
  const int bid = blockIdx.x;
  const DequantJob job = jobs[find_dequant_job (jobs, njobs, bid)];
  const int t = bid - job.tile_base;
  const int ty = t / job.tiles_x, tx = t - ty * job.tiles_x;
  const int x = tx * kDqTW + 4 * (threadIdx.x & 15), y = ty * kDqTH + (threadIdx.x >> 4);
  if (x >= job.w || y >= job.h)
    return;
  const int n = min (4, job.w - x);
  T v[4] = { 0, 0, 0, 0 };
  if (job.src) {
    const size_t s = (size_t) y * job.w + x;
    for (int e = 0; e < n; e++) {
      const int32_t q = job.src_bytes == 1 ? (int32_t) gload < int8_t > ((const int8_t *) job.src + s + e)
          : job.src_bytes == 2 ? (int32_t) gload < int16_t > ((const int16_t *) job.src + s + e)
          : gload < int32_t > ((const int32_t *) job.src + s + e);
      v[e] = dequant_one < T, ARITH > (q, job.factor, job.offset);
    }
  }
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
