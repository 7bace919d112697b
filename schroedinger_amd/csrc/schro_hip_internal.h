// schro_hip_internal.h -- shared between the HIP translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <cstddef>
#include <vector>

#include "schro_hip.h"

// The device-free build of the host code for the sanitizers (schro_hip_dry.h: test infrastructure, never shipped): the
// HIP runtime's entry points become host stand-ins and launches are dropped.
#ifdef SCHRO_HIP_DRY
#include "schro_hip_dry.h"
#endif

// Switches that force one of the library's kernels where another would be chosen (the tests' second
// formulations, A/B runs) and the measured-slower forms they select live in the EXPERIMENTS build only
// (libschro_hip_exp.so: make exp, -DSCHRO_HIP_EXPERIMENTS; tests/test_gpu_experiments.py runs it in child
// processes).  The product library reads no SCHRO_HIP_* variable but SCHRO_HIP_DEBUG.
#ifdef SCHRO_HIP_EXPERIMENTS
#define SCHRO_ENV(name) getenv (name)
#else
#define SCHRO_ENV(name) ((const char *) nullptr)
#endif

namespace schro {

int set_error (int code, const char *fmt, ...);
// the same, but never aborts: an answer the caller routes on (SCHRO_HIP_ENEEDS_RESIDUAL), not a trapped assertion
int set_status (int code, const char *fmt, ...);

static inline int
div_up (int a, int b)
{
  return (a + b - 1) / b;
}

static inline size_t
round_up (size_t a, size_t b)
{
  return (a + b - 1) / b * b;
}

#define SCHRO_HIP_CHECK(expr)                                                  \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess)                                                      \
      return schro::set_error (SCHRO_HIP_EDEVICE, "%s:%d: %s -> %s", __FILE__, \
          __LINE__, #expr, hipGetErrorString (_e));                            \
  } while (0)

#define SCHRO_HIP_REQUIRE(cond, ...)                                           \
  do {                                                                         \
    if (!(cond))                                                               \
      return schro::set_error (SCHRO_HIP_EINVAL, __VA_ARGS__);                 \
  } while (0)

// Every kernel launch of the library goes through SCHRO_LAUNCH.  While a ProfileScope of a
// profiling context is open (schro_hip_profile_enable), the launch carries a start / stop event pair
// (hipExtLaunchKernelGGL): the events take the kernel's own begin and end from the dispatch's completion
// signal -- the duration rocprofv3 reports -- and put nothing else on the queue.  (r02 bracketed every
// launch with two hipEventRecord: 8 us of barrier packets per launch, 0.12 ms per profiled bench step.)
bool profile_launch_events (hipEvent_t * start, hipEvent_t * stop);

#ifdef SCHRO_HIP_DRY
// (the grid and the block are still evaluated: a launch geometry computed from a bad table shows up in UBSAN)
#define SCHRO_LAUNCH(kernel, grid, block, shmem, stream, ...)                                      \
  do {                                                                                             \
    const dim3 g_ = (grid), b_ = (block);                                                          \
    (void) g_; (void) b_; (void) (shmem); (void) (stream);                                         \
  } while (0)
#else
#define SCHRO_LAUNCH(kernel, grid, block, shmem, stream, ...)                                      \
  do {                                                                                             \
    hipEvent_t pa_, pb_;                                                                           \
    if (schro::profile_launch_events (&pa_, &pb_))                                                 \
      hipExtLaunchKernelGGL (kernel, grid, block, shmem, stream, pa_, pb_, 0, __VA_ARGS__);       \
    else                                                                                           \
      hipLaunchKernelGGL (kernel, grid, block, shmem, stream, __VA_ARGS__);                        \
  } while (0)
#endif

// ---- job tables handed to the kernels (device memory, one per launch) ------

// One (plane, level) of the inverse wavelet.
struct IwtJob {
  const void *sb[4];            // LL, HL, LH, HH: element (0,0) of each sub-band
  int sb_stride[4];             // bytes between sub-band rows
  void *dst;
  int dst_stride;
  int w, h;                     // output size of this level (sub-bands are w/2 x h/2)
  int tiles_x;
  int tile_base;                // first block id of this job
  int flags;                    // bit0: all sources 8-byte aligned, bit1: dst 16-byte aligned
  int pad;
  // chain form (iiwt_reg.hip, r04): the job whose output is this job's LL band -- rows of output per tile row
  // (0: none, the coarsest level), its tile grid, its first counter -- this job's own first counter (-1: nobody
  // reads its output in the launch: level 0), and its tile form
  int dep_rows2, dep_tiles_y, dep_tiles_x, dep_ctr;
  int ctr, small;
  // combine form (r04, level 0 of the register kernel): dst is the u8 PICTURE, out_w x out_h inside the
  // iwt-padded w x h; pred: the OBMC prediction to add (u8), NULL: + 128 (a picture without references)
  const uint8_t *pred;
  int pred_stride;
  int out_w, out_h;
  int pad2;
};

// r05: the three-level s32 Haar transform of a 4:2:2 picture with the v210 copy-out as its epilogue (iiwt_haar.hip)
struct HaarPackJob {
  const void *src[3];           // the coefficient planes (Y, U, V), in-place sub-band layout
  int src_stride[3];
  int w, h;                     // luma transform size = picture size (chroma: w / 2 x h)
  uint8_t *dst;                 // v210 rows: 16 bytes per 6 pixels
  int dst_stride;
  int tiles_x;                  // workgroups per strip of 8 rows
  int tile_base;
};

struct ConvertJob {
  const void *src;
  uint8_t *dst;
  int src_stride, dst_stride;
  int w, h;
  int tiles_x;
  int tile_base;
  const uint8_t *pred;          // r04: NULL: + 128 (offsetconvert); else the prediction to add (rrshift6_add's last steps)
  int pred_stride;
  int pad;
};

struct PackJob {
  const uint8_t *src[3];
  uint8_t *dst;
  int src_stride[3];
  int dst_stride;
  int sw, sh;                   // source luma size
  int hs, vs;                   // source chroma shifts
  int w, h;                     // packed picture size
  int format;
  int src_bpp;                  // v210 only: 1, 2 or 4
  int tiles_x;
  int tile_base;
};

struct UpsampleJob {
  const uint8_t *src;
  uint8_t *dst;
  int src_stride, dst_stride;
  int w, h;
  int tiles_x;
  int tile_base;
  // r04: the V plane of a (U, V) pair whose half-pel samples are stored byte-interleaved (see the
  // half-pel layout below); NULL: one plane
  const uint8_t *src_b;
  int src_b_stride;
  int pad;
};

struct ObmcJob {
  const uint8_t *mvs;
  const uint8_t *ref[2];
  const void *residual;
  uint8_t *out;
  int ref_stride[2];
  int residual_stride;
  int out_stride;
  int w, h;
  int nbx, nby;                 // x_num_blocks, y_num_blocks
  int xblen, yblen, xbsep, ybsep, xoff, yoff;   // this component's geometry
  int max_x_blocks, max_y_blocks;       // interior-block limits (schromotion8.c:794-797)
  int mv_shift_x, mv_shift_y;   // chroma MV scaling (0 for luma)
  int prec, wbits, w1, w2;
  int comp;
  int res_bpp;
  int tiles_x;
  int tile_base;
  // item kernel: everything that depends only on the plane's block geometry, worked out
  // on the host (obmc_item_geometry) instead of in every workgroup's prologue; m_* are
  // ceil (2^32 / d) for mdiv ()
  // (ref_ps, ref_cb, r04: how a sample of this component lies in its half-pel planes -- a sample is
  // 1 << ref_ps bytes wide and this component is byte ref_cb of it: 0, 0 one component per image; 1, c the
  // U / V samples of a picture interleaved, see the half-pel layout below.  The struct stays 256 bytes: the
  // kernels keep a copy in scalar registers.)
  int nseg, ref_ps, lpi, ref_cb, ipw, chunk_cap;
  uint32_t m_tiles_x, m_xbsep, m_ybsep, m_nseg, m_lpi;
  uint32_t m_xramp, m_yramp;    // ceil (2^32 / (2 * offset - 1)): get_ramp's division (schromotion.c:40-49)
  int out_s16;                  // r06: `out` is an s16 plane that receives (acc - 8160) >> 6 = the prediction - 128
                                // (orc_rrshift6_s16_ip_2d: schro_motion_render's add = FALSE, schro_motion_render_cuda's dest);
                                // no residual.  obmc.hip's kernels only
  unsigned long long *stamps;   // scratch runs only (SCHRO_HIP_OBMC_STAMPS): per-workgroup phase stamps
  // row kernel: the second plane of a job.  The U and V planes of a picture have the same
  // blocks, vectors and sample windows, so one workgroup decodes a tile's blocks once and
  // predicts both planes (nplanes == 2); everything not listed here is shared with plane A
  int nplanes;
  int comp_b;
  const uint8_t *ref_b[2];
  const void *residual_b;
  uint8_t *out_b;
  int residual_stride_b, out_stride_b;
};
static_assert (sizeof (ObmcJob) == 256, "ObmcJob: four 64-byte scalar loads");

// One picture's slices (lowdelay.hip).
struct SliceJob {
  const uint8_t *data;
  uint32_t data_bytes;
  int pad;
  void *comp[3];
  int stride[3];
  int pad2;
};

// What every slice of a launch shares (passed by value).
struct SliceParams {
  int depth;
  int iwt_lw, iwt_lh, iwt_cw, iwt_ch;
  int nh, nv;
  int n_bytes, remainder, denom;        // slice_bytes_num / _denom split, schrolowdelay.c:601-602
  int quant_matrix[SCHRO_HIP_LIMIT_SUBBANDS];
  int run_cap;                          // slice_run_kernel: staging words per lane (launch_slices sets it)
};

struct DcJob {
  void *data;
  int stride;
  int w, h;
  int pad;
};

// One codeblock of the core-syntax dequantisation (dequant.hip).
struct DequantJob {
  void *dst;                    // first sample of the codeblock in the coefficient frame
  const void *src;              // its quantised values (row-major, tight); NULL: zero codeblock
  int dst_stride;
  int w, h;
  int src_bytes;
  uint32_t factor, offset;
  int tiles_x;
  int tile_base;
};

// r04, dequantisation plans: what is fixed per picture geometry (device-resident) ...
struct DequantGeo {
  int dst_offset;               // bytes from the plane's base to the codeblock's first sample
  int dst_stride;
  int w, h;
  int tiles_x;
  int tile_base;
  int plane;                    // index into the run's DequantPlaneDyn table
  int rec;                      // index into the run's SchroHipCodeblock records
};
// ... and per run and plane (the records themselves go up as they are)
struct DequantPlaneDyn {
  void *dst;
  const void *values;
  int is_intra;
  int pad;
};

constexpr int kMaxJobs = 256;

// XCD-aware workgroup order.  The dispatcher deals workgroups round-robin over
// the 8 XCDs (each with its own 4 MiB L2), so neighbouring tiles -- which share
// lifting halos and reference windows -- would land on 8 different L2s and each
// would fetch its own copy.  This bijection gives every XCD one contiguous run
// of tile ids instead (speed only; any placement is correct).
#ifdef __HIPCC__
// Device pointers reach the kernels inside job structs, so the compiler only knows them
// as generic pointers and would emit flat_load / flat_store: those count on vmcnt AND
// lgkmcnt and return out of order, which forces s_waitcnt 0 before any LDS result is
// used and serialises a software-pipelined load with the work it should overlap.
// gload / gstore say "this is global memory" at the access (global_load / global_store).
// Native vector types only: a class type such as uint2 would be copied through a generic
// reference and fall back to flat.
#define SCHRO_GLOBAL __attribute__ ((address_space (1)))
typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
typedef uint32_t u32_u __attribute__ ((aligned (1)));         // byte-aligned forms
typedef u32x2 u32x2_u __attribute__ ((aligned (1)));
typedef u32x4 u32x4_u __attribute__ ((aligned (1)));

template < typename V, typename P > __device__ __forceinline__ V
gload (const P * p)
{
  return *(const SCHRO_GLOBAL V *) p;
}

template < typename V, typename P > __device__ __forceinline__ void
gstore (P * p, V v)
{
  *(SCHRO_GLOBAL V *) p = v;
}

// n / d for 1 <= d <= 1024 and 0 <= n < 2^22 without the ~35-instruction integer division
// sequence (there is no hardware divide): one v_mul_hi_u32 by ceil (2^32 / d).  Tile and
// block geometry (tiles per row, block separation, lanes per item ...) is divided by in
// every workgroup's prologue; the item kernel spent ~500 instructions per wave there.
struct DivMagic {
  uint32_t m[1025];
};
constexpr DivMagic
make_div_magic ()
{
  DivMagic t = { };
  for (uint32_t d = 2; d <= 1024; d++)
    t.m[d] = (uint32_t) ((0x100000000ull + d - 1) / d);
  return t;
}
static __device__ __constant__ DivMagic kDivMagic = make_div_magic ();

__device__ __forceinline__ int
mdiv (int n, int d, uint32_t m)
{
  return d == 1 ? n : (int) __umulhi ((uint32_t) n, m);
}

__device__ __forceinline__ int
fdiv (int n, int d)
{
  if (d > 1024)                 // outside the table (no geometry of this path gets here)
    return n / d;
  return d == 1 ? n : (int) __umulhi ((uint32_t) n, kDivMagic.m[d]);
}

// n / d with the host-side magic m = div_magic (d): no table load in the kernel
__host__ __device__ inline uint32_t
div_magic (int d)
{
  return d <= 1 ? 0u : (uint32_t) ((0x100000000ull + (uint32_t) d - 1) / (uint32_t) d);
}

// Half-pel images (include/schro_hip.h), r03 layout.  The four planes of the reference's upsampled
// frame (integer, h-half, v-half, hv-half: plane = (X & 1) + 2 * (Y & 1) of half-pel sample (X, Y))
// are kept apart, so a block row's prediction samples are CONTIGUOUS bytes of one plane (no
// even / odd split in the kernel) and a tap the block's phase does not use is never fetched.
// A plane row is stored as 32-byte chunks that advance by 16 columns -- every column sits in two
// chunks -- so any run of up to 17 bytes starts inside some chunk and ends inside the same one: a
// lane fetches its row with ONE byte-aligned load, no alignment or phase-select instructions.  One
// 128-byte line = the same chunk of 4 consecutive rows of one plane; the lines of the four planes
// of a (band of 4 rows, chunk) are adjacent (512 bytes), so the other taps of a window are at
// +-128 / +-256 bytes.  kHpApron replicated columns lie in front of column 0 and behind the last
// one -- get_block clamps a block's origin to 32 pixels outside the picture (schromotion8.c:
// 329-330) -- with the reference's sources (schroframe.c:2012-2029: planes 0 and 1 repeat plane
// 0's edge, planes 2 and 3 plane 2's), which is the clamp of the half-pel column to [0, 2w - 2];
// rows are clamped by the kernels.  `stride` = bytes per band of 4 rows = chunks * 512.
//
// r04 -- chroma as (U, V) pairs.  The U and V planes of a 4:2:0 / 4:2:2 picture have the same blocks,
// vectors and sample windows; stored apart, every window was fetched twice (a 6 x 6 chroma window
// touches 2.25 lines per tap and plane).  A PAIR image is the same layout over samples of two bytes
// (U, V): byte column 2 * (column + kHpApron) + c of the plane row, chunks of 32 bytes = 16 byte
// columns of advance = 8 samples, aprons of 2 * kHpApron bytes.  A block row of up to 8 samples plus
// its X + 1 tap is a run of <= 18 bytes that starts at an even byte <= 14 of its chunk: the same one
// load per tap as for a luma row, and it brings both components.  `ps` (sample shift: 0 plane, 1 pair)
// and `cb` (byte of the sample) below select the form; widths in hp_chunks () are SAMPLES.
constexpr int kHpApron = 32;
constexpr int kHpBandRows = 4;

__host__ __device__ __forceinline__ int
hp_chunks (int w, int ps = 0)
{
  return (((w + 2 * kHpApron) << ps) + 15) / 16 + 1;
}

// padded column xp (= plane column + kHpApron) inside its band: chunk xp >> 4, byte xp & 15 (the
// same column is also byte 16 + (xp & 15) of chunk (xp >> 4) - 1)
__host__ __device__ __forceinline__ size_t
hp_col_offset (int xp)
{
  return (size_t) (xp >> 4) * 512 + (size_t) (xp & 15);
}

__host__ __device__ __forceinline__ size_t
hp_row_offset (int y, int stride)
{
  return (size_t) (y >> 2) * (size_t) stride + (size_t) ((y & 3) * 32);
}

// byte offset of half-pel sample (X, Y), 0 <= X <= 2w - 1, 0 <= Y <= 2h - 1 (pair images: of its
// component cb)
__host__ __device__ __forceinline__ size_t
hp_offset (int X, int Y, int stride, int ps = 0, int cb = 0)
{
  return hp_row_offset (Y >> 1, stride) + hp_col_offset ((((X >> 1) + kHpApron) << ps) + cb) + (size_t) (((X & 1) + 2 * (Y & 1)) * 128);
}

__device__ __forceinline__ int
xcd_tile_id (int bid, int nblocks)
{
  constexpr int kXcd = 8;
  const int q = nblocks / kXcd, r = nblocks % kXcd;
  const int x = bid % kXcd, k = bid / kXcd;
  return x * q + (x < r ? x : r) + k;
}

// Which job owns workgroup tile `bid`: the jobs' tile_base values ascend from 0, so the
// answer is (number of jobs with tile_base <= bid) - 1.  The lanes of the wave probe 64
// jobs at a time (one memory round trip instead of a chain of dependent scalar loads,
// which cost ~6.6k cycles per workgroup for the 24 planes of 8 pictures).
template < typename JOB >
__device__ __forceinline__ int
find_job (const JOB * jobs, int njobs, int bid)
{
  const int lane = threadIdx.x & 63;
  int n = 0;
  for (int base = 0; base < njobs; base += 64) {
    const int idx = base + lane;
    const bool le = idx < njobs && gload < int > (&jobs[idx].tile_base) <= bid;
    n += __popcll (__ballot (le));
  }
  return __builtin_amdgcn_readfirstlane (n - 1);
}
#endif

// launchers (one per .hip file)
int launch_iiwt_level (hipStream_t stream, const IwtJob * d_jobs, int njobs,
    int total_tiles, int filter, int bpp);

// fused finest levels (iiwt.hip): one launch runs levels nl-1 .. 0
size_t iiwt_fused_job_size (void);
int iiwt_fused_max_levels (int filter, int bpp);
void iiwt_fused_job_fill (void *job, const void *src, int src_stride, int bpp, int nl,
    const void *ll, int ll_stride, void *dst, int dst_stride, int w, int h, int tiles_x,
    int tile_base);
int launch_iiwt_fused (hipStream_t stream, const void *d_jobs, int njobs, int total_tiles,
    int filter, int bpp, int nl);
// element-wise s32 Haar level (iiwt_haar.hip)
bool iiwt_haar_supported (int filter, int bpp);
bool iiwt_haar_job_ok (const IwtJob & j);
void iiwt_haar_geometry (int *cols, int *rows);
int launch_iiwt_haar (hipStream_t stream, const IwtJob * d_jobs, int njobs, int total_tiles, int filter);
// ... all three levels of a depth-3 transform in one pass (r03)
bool iiwt_haar3_job_ok (const void *src, int src_stride, const void *dst, int dst_stride, int w, int h);
void iiwt_haar3_geometry (int *blocks_x, int *blocks_y);
int launch_iiwt_haar3 (hipStream_t stream, const IwtJob * d_jobs, int njobs, int total_tiles, int filter);
// register form of one level (iiwt_reg.hip): s16, filters with a small lifting halo
bool iiwt_reg_supported (int filter, int bpp);
void iiwt_reg_geometry (int filter, int small, int *useful_cols, int *useful_row_pairs,
    int *min_row_pairs);
int launch_iiwt_reg (hipStream_t stream, const IwtJob * d_jobs, int njobs, int total_tiles,
    int filter, int small, int combine);
int launch_iiwt_chain (hipStream_t stream, const IwtJob * d_jobs, const uint32_t * d_order, int n_tiles, uint32_t * ctrl,
    uint32_t run, uint32_t * gave_up, uint32_t epoch, int filter);
void iiwt_tile_geometry (int filter, int bpp, int *useful_cols,
    int *useful_row_pairs);
int launch_convert (hipStream_t stream, const ConvertJob * d_jobs, int njobs,
    int total_tiles, int bpp);
void convert_tile_geometry (int *tw, int *th);
int launch_pack (hipStream_t stream, const PackJob * d_jobs, int njobs, int total_tiles);
int launch_shift_right (hipStream_t stream, const ConvertJob * d_jobs, int njobs, int total_tiles, int bpp, int shift);
int launch_add (hipStream_t stream, const ConvertJob * d_jobs, int njobs, int total_tiles, int src_bpp);
void pack_tile_geometry (int *groups_x, int *rows);
// persist_grid > 0 (experiments build): that many persistent workgroups (all jobs of one form: pair images or planes)
int launch_upsample (hipStream_t stream, const UpsampleJob * d_jobs,
    int njobs, int total_tiles, int persist_grid);
void upsample_tile_geometry (int *tw, int *th);
int launch_slices (hipStream_t stream, const SliceJob * d_jobs, int njobs, const SliceParams & P, int bpp, int arith,
    bool aligned16);
bool dc_skew_ok (const DcJob * jobs, int njobs, int bpp);
// edge != NULL: dc_skew_kernel with that hand-over buffer (edge_pitch samples per strip, njobs x strips of
// them, tagged with `epoch`); NULL: dc_predict_kernel
int launch_dc_predict (hipStream_t stream, const DcJob * d_jobs, int njobs, int max_rows, int bpp,
    unsigned long long *edge, int edge_pitch, uint32_t epoch, uint32_t * gave_up);
// the hand-over buffer of the selected queue for a launch of njobs bands of at most max_rows x max_w
int dc_edge_for (SchroHipContext * ctx, int njobs, int max_rows, int max_w, unsigned long long **edge,
    int *edge_pitch, uint32_t * epoch);
// non-zero (the launch's epoch) once a dc_skew_kernel strip or a chain-wavelet tile has given up waiting (a device fault
// of the launch): reported by the next call that looks, enqueueing or synchronising
int dc_gave_up (SchroHipContext * ctx);
// r06: the prediction_only OBMC batches whose predictions did not fit 8 bits -- a per-picture ROUTING answer
// (SCHRO_HIP_ENEEDS_RESIDUAL), looked for by the synchronising calls only (stage completion, schro_hip_synchronize,
// schro_hip_queue_synchronize) and by schro_hip_obmc_overflowed: an unrelated enqueue is never refused for it
int pred_overflow_poll (SchroHipContext * ctx);
// before batch `epoch` is given ring word epoch % kOvfRing: the word's previous owner has finished and its answer is kept
int pred_overflow_claim (SchroHipContext * ctx, uint32_t epoch);
}
// (plane_lowdelay.cpp, beside the plan's other entry points; not part of the public header)
extern "C" bool schro_hip_dequant_plan_matches (const SchroHipDequantPlan * plan, const SchroHipDequantPlane * planes, int nplanes, int bpp, int arith);
namespace schro {
// iiwt_haar.hip, r05
bool iiwt_haar3_v210_ok (const HaarPackJob & j);
int iiwt_haar3_v210_strip_width ();
int launch_iiwt_haar3_v210 (hipStream_t stream, const HaarPackJob * d_jobs, int njobs, int total_tiles, int filter);
int launch_dequant (hipStream_t stream, const DequantJob * d_jobs, int njobs, int total_tiles, int bpp, int arith);
int launch_dequant_plan (hipStream_t stream, const DequantGeo * d_geo, int njobs, int total_tiles,
    const SchroHipCodeblock * d_recs, const DequantPlaneDyn * d_planes, int bpp, int arith);
void dequant_tile_geometry (int *tw, int *th);
int launch_table_copy (hipStream_t stream, void *dst, const void *src, size_t bytes);
// schro_table_quant[i] and schro_table_offset_1_2[i] (intra) / _3_8[i] (inter)
void dequant_tables (int quant_index, int is_intra, uint32_t * factor, uint32_t * offset);
// overflow: NULL, or (prediction_only launches) the word a prediction that does not fit 8 bits is reported in
int launch_obmc (hipStream_t stream, const ObmcJob * d_jobs, int njobs,
    int total_tiles, int prec, int variant, const uint32_t * d_order, uint32_t * overflow);
// row kernels (obmc_row*.hip): prediction dwords per block row and segment, *ns = segments per block row (1, 2);
// 0 = not their case
int obmc_row_form (const ObmcJob & job, bool uv, int *ns);
// is there a kernel of `np` planes per job (1, 2; 3: (U, V) pairs from pair images) for this precision and form?
bool obmc_row_has_kernel (int prec, int nd, int np, int ns, bool weighted);
// obmc_strip.hip (r05): the register-accumulator form for the 12 / 8 block set
bool obmc_strip_ok (const ObmcJob & j);
void obmc_strip_tiles (const ObmcJob & j, int seg_rows, int *strips, int *segs);
int launch_obmc_strip (hipStream_t stream, const ObmcJob * d_jobs, int njobs, int total_items, int seg_rows, bool nores, uint32_t * overflow,
    int cus);
int obmc_row_tile_width (bool uv);
int obmc_row_tile_height ();
int launch_obmc_row (hipStream_t stream, const ObmcJob * d_jobs, int njobs, int total_tiles, int prec, int nd, int ns,
    int planes_per_job, const uint32_t * d_order, uint32_t * overflow, const uint32_t * d_wtabs, bool weighted);
// the weight table of a job's block geometry as the row kernels copy it into LDS (ObmcJob::ipw: its index in d_wtabs)
// words 1 .. 3 of tile (tx, ty)'s record in a row launch's order table (word 0: job << 16 | tile)
void obmc_row_tile_record (const ObmcJob & job, bool uv, int ns, int tx, int ty, uint32_t * rec);
int obmc_row_weight_words (int nd, int ns);
void obmc_row_weight_table (const ObmcJob & job, int nd, int ns, bool uv, uint32_t * out);
// fills the item-kernel geometry fields of a job (obmc.hip)
void obmc_item_geometry (ObmcJob * job);
void obmc_tiles (int variant, int w, int h, int xoff, int *tiles_x, int *tiles_y);

}                               // namespace schro

// ---- the context --------------------------------------------------------------

struct SchroHipContext {
  int device;
  SchroHipMemoryDomain *domain; // the SchroMemoryDomain-shaped handle of this context
  // Two in-order queues (HIP streams) per context: the pixel path of one batch of pictures is
  // HBM-bound in the inverse wavelet and issue-bound in OBMC, so a decoder that runs batch k's
  // OBMC on one queue and batch k+1's wavelet + upsample on the other keeps both busy.  Calls
  // go to the selected queue; `stream` is always streams[cur].  Job-table slots, tile-order
  // slots and the wavelet's scratch are per queue, so a launch on one queue never has its
  // tables rewritten by a copy enqueued on the other.
  // r03: four queues -- 0 and 1 for kernels as before, 2 and 3 by convention the host-to-device and
  // the device-to-host copy queue (SCHRO_HIP_QUEUE_H2D / _D2H): copies from / to pinned host memory
  // enqueued there run on the DMA engines beside the kernels of the other queues, ordered by marks.
  static constexpr int kQueues = 4;
  hipStream_t streams[kQueues];
  hipEvent_t queue_ev[kQueues];
  static constexpr int kMarks = 16;
  hipEvent_t marks[kMarks];     // schro_hip_queue_mark / _wait_mark, created on first use
  int cur;
  hipStream_t stream;
  hipEvent_t ev_begin, ev_end;
  bool stage_complete;          // frame-layer stage calls wait for the selected queue before they return (default)

  // size-keyed allocation cache (schrodomain.c:58-137 semantics)
  struct Slot {
    void *ptr;
    size_t size;
    bool in_use;
  };
  std::vector < Slot > slots;
  size_t domain_bytes;

  // Job tables on the device.  A decoder calls the batch entry points with the same few
  // sets of planes over and over (its frame pool), so tables are kept in a small
  // content-addressed cache: a repeated table costs a hash + memcmp on the host and no
  // copy at all; a new one replaces the least recently used slot (one async copy from the
  // slot's pinned mirror, ordered on the stream behind the kernels that still read it).
  struct ArgSlot {
    uint64_t hash;
    size_t bytes;               // 0: empty
    uint64_t last_use;
    hipEvent_t copied;          // the slot's last host -> device copy
    bool copy_pending;
  };
  static constexpr int kArgSlots = 256;        // kArgSlots / kQueues per queue
  static constexpr size_t kArgSlotBytes = 64u << 10;   // >= kMaxJobs OBMC jobs (static_assert in plane_obmc.cpp)
  char *h_args;                 // kArgSlots pinned mirrors
  char *d_args;
  ArgSlot arg_slots[kArgSlots];
  uint64_t arg_clock;

  // optional per-kernel event profiling
  bool profile;
  struct EvPair {
    hipEvent_t a, b;
    int cls;
  };
  std::vector < EvPair > ev_pool;
  size_t ev_used;

  // OBMC tile orders (plane_obmc.cpp obmc_tile_order): device tables of job << 16 | tile, cached by
  // the geometry and references of the launch they were built for
  struct OrderSlot {
    uint64_t hash;
    uint32_t *d;
    uint32_t *h;                // pinned mirror the table is uploaded from
    size_t cap, count;
    uint64_t last_use;
    hipEvent_t copied;          // the slot's last upload
    bool copy_pending;
  };
  static constexpr int kOrderSlots = 16;        // kOrderSlots / kQueues per queue
  OrderSlot order_slots[kOrderSlots];

  // the register wavelet's chain form (plane_iiwt.cpp iiwt_chain): tile orders cached by the batch's geometry, and per
  // queue the ticket + counters its launches synchronise through (left zero by every launch)
  struct ChainSlot {
    uint64_t hash;
    uint32_t *d;
    size_t cap, count;
    uint64_t last_use;
  };
  static constexpr int kChainSlots = 16;        // kChainSlots / kQueues per queue
  ChainSlot chain_slots[kChainSlots];
  uint32_t *chain_ctrl[kQueues];
  size_t chain_ctrl_words[kQueues];
  uint64_t chain_ctrl_hash[kQueues];    // the geometry the counters count for ...
  uint32_t chain_runs[kQueues];         // ... and how many launches of it they have counted

  // grow-only scratch for intermediate LL bands, one per queue
  void *scratch_q[kQueues];
  size_t scratch_size_q[kQueues];
  void *&scratch_ref () { return scratch_q[cur]; }
  // Job tables too large for a table slot (the codeblocks of a whole batch of pictures): four
  // grow-only pinned mirrors + device buffers per queue, used in turn (the host runs up to three
  // batches ahead of the device), never cached (a decoder's codeblock tables differ from picture to
  // picture)
  struct BigTable {
    char *h, *d;
    size_t cap;
    hipEvent_t copied;
    bool pending;
  };
  static constexpr int kBigTables = 4;
  BigTable big_q[kQueues][kBigTables];
  int big_turn[kQueues];
  // dc_skew_kernel's hand-over buffers (one per queue: two launches in flight never share one) and
  // the launch counter their samples are tagged with
  void *dc_edge_q[kQueues];
  size_t dc_edge_size_q[kQueues];
  uint32_t dc_epoch;
  uint32_t *dc_gave_up;         // pinned host words (16): [0] the epoch of a DC launch whose strip gave up, [1] of a chain
                                // wavelet launch, [4 .. 15] r05: a ring of flags, one per prediction_only OBMC batch in turn
  // r05: prediction_only OBMC batches are numbered (1, 2, ...); batch e raises word 4 + e % kOvfRing when one of its
  // predictions does not fit 8 bits, and ovf_epoch[] remembers which batch a ring word was last handed to
  // r05: the frame layer's dequantisation plan (schro_hipframe_dequantise): one geometry at a time
  struct SchroHipDequantPlan *frame_dq_plan;
  void *dq_stage_q[kQueues];    // staging of host-side quantised values, one per queue (in-order reuse)
  size_t dq_stage_size_q[kQueues];
  // r06: the pixel planes of schro_hip_iiwt_pack_v210_batch's two-pass route, one grow-only block per queue (in-order
  // reuse: the next call's transform writes them behind this call's pack on the same queue) -- no allocation, no wait per call
  void *pack_tmp_q[kQueues];
  size_t pack_tmp_size_q[kQueues];
  static constexpr int kOvfRing = 12;
  uint32_t pred_epoch;
  uint32_t ovf_epoch[kOvfRing];         // the batch that owns ring word k (0: nobody)
  hipEvent_t ovf_ev[kOvfRing];          // ... recorded behind its launches (r06): the word is read and handed on only once it has fired
  // r06: batches found raised, oldest first -- not yet named by a synchronising call's status / not yet fetched by
  // schro_hip_obmc_overflowed (nothing is dropped: a word is cleared only into these lists)
  std::vector < uint32_t > ovf_unannounced, ovf_unfetched;
  int cus;                      // compute units of the device (launch shaping)
};

namespace schro {
// returns a device pointer to a copy of [host, host+bytes), valid for launches enqueued
// on the context's stream before the next 63 distinct tables
int push_args (SchroHipContext * ctx, const void *host, size_t bytes,
    void **dev);
int ensure_scratch (SchroHipContext * ctx, size_t bytes);
// a device copy of a job table of any size, valid for the launches enqueued on the context's stream
// before the fourth call from now on this queue
int push_big_table (SchroHipContext * ctx, const void *host, size_t bytes, void **dev);
// ... built in place: *host is the pinned mirror to fill, big_table_commit sends it
int big_table_begin (SchroHipContext * ctx, size_t bytes, void **host, void **dev);
int big_table_commit (SchroHipContext * ctx, size_t bytes);
// rows of row_bytes bytes, host <-> device or device -> device, enqueued on the selected queue (context.cpp)
int copy_2d_async (SchroHipContext * ctx, void *dst, int dst_stride, const void *src, int src_stride, int row_bytes,
    int height, hipMemcpyKind kind);
// v216 / ARGB / AY64 (plane_frameops.cpp)
bool is_wide_format (int format);
// the launches made while a scope is open are timed under its kernel class (when profiling is on)
struct ProfileScope {
  ProfileScope (SchroHipContext * c, int cls);
  ~ProfileScope ();
};
}
