// plane_frameops.cpp -- plane layer: intra convert, packed copy-out (YUYV / UYVY / AYUV / v210 / v216 / ARGB / AY64), the > 8-bit
// shift, the half-pel upsample and the helpers around its tiled planes (frameops.hip).

#include "schro_hip_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

using namespace schro;

extern "C" {

int
schro_hip_convert_u8_batch (SchroHipContext * ctx, const SchroHipConvertPlane * planes,
    int nplanes, int bpp)
{
  SCHRO_HIP_REQUIRE (ctx && planes && nplanes > 0 && nplanes <= kMaxJobs,
      "convert_batch: bad arguments");
  SCHRO_HIP_REQUIRE (bpp == 2 || bpp == 4, "convert_batch: bpp must be 2 or 4");
  (void) hipSetDevice (ctx->device);
  int tw, th;
  convert_tile_geometry (&tw, &th);
  std::vector < ConvertJob > jobs (nplanes);
  int tile_base = 0;
  for (int p = 0; p < nplanes; p++) {
    const SchroHipConvertPlane & pl = planes[p];
    SCHRO_HIP_REQUIRE (pl.src && pl.dst && pl.width > 0 && pl.height > 0,
        "convert_batch: plane %d invalid", p);
    ConvertJob & j = jobs[p];
    j.src = pl.src;
    j.dst = pl.dst;
    j.src_stride = pl.src_stride;
    j.dst_stride = pl.dst_stride;
    j.w = pl.width;
    j.h = pl.height;
    j.tiles_x = div_up (pl.width, tw);
    j.tile_base = tile_base;
    tile_base += j.tiles_x * div_up (pl.height, th);
  }
  void *d_jobs;
  int r = push_args (ctx, jobs.data (), sizeof (ConvertJob) * nplanes, &d_jobs);
  if (r)
    return r;
  ProfileScope ps (ctx, SCHRO_HIP_KERNEL_CONVERT);
  return launch_convert (ctx->stream, (const ConvertJob *) d_jobs, nplanes, tile_base, bpp);
}

// dst (s16) += src (s16 | u8): schro_frame_add / schro_gpuframe_add on planes (schroframe.c:1082-1135,
// schrogpuframe.c:257-306); width x height = the planes' common size
int
schro_hip_add_batch (SchroHipContext * ctx, const SchroHipConvertPlane * planes, int nplanes, int src_bytes_per_sample)
{
  SCHRO_HIP_REQUIRE (ctx && planes && nplanes > 0 && nplanes <= kMaxJobs, "add_batch: bad arguments");
  SCHRO_HIP_REQUIRE (src_bytes_per_sample == 1 || src_bytes_per_sample == 2, "add_batch: the source is u8 or s16");
  (void) hipSetDevice (ctx->device);
  int tw, th;
  convert_tile_geometry (&tw, &th);
  std::vector < ConvertJob > jobs (nplanes);
  int tile_base = 0;
  for (int p = 0; p < nplanes; p++) {
    const SchroHipConvertPlane & pl = planes[p];
    SCHRO_HIP_REQUIRE (pl.src && pl.dst && pl.width > 0 && pl.height > 0 && pl.dst_stride >= 2 * pl.width
        && pl.src_stride >= src_bytes_per_sample * pl.width && ((uintptr_t) pl.dst | (uintptr_t) pl.dst_stride) % 2 == 0
        && ((uintptr_t) pl.src | (uintptr_t) pl.src_stride) % src_bytes_per_sample == 0, "add_batch: plane %d invalid", p);
    ConvertJob & j = jobs[p];
    memset (&j, 0, sizeof (j));
    j.src = pl.src;
    j.dst = pl.dst;
    j.src_stride = pl.src_stride;
    j.dst_stride = pl.dst_stride;
    j.w = pl.width;
    j.h = pl.height;
    j.tiles_x = div_up (pl.width, tw);
    j.tile_base = tile_base;
    tile_base += j.tiles_x * div_up (pl.height, th);
  }
  void *d_jobs;
  int r = push_args (ctx, jobs.data (), sizeof (ConvertJob) * nplanes, &d_jobs);
  if (r)
    return r;
  ProfileScope ps (ctx, SCHRO_HIP_KERNEL_CONVERT);
  return launch_add (ctx->stream, (const ConvertJob *) d_jobs, nplanes, tile_base, src_bytes_per_sample);
}

}                               // extern "C"

namespace schro {
bool
is_wide_format (int f)
{
  return f == SCHRO_HIP_FORMAT_v216 || f == SCHRO_HIP_FORMAT_ARGB || f == SCHRO_HIP_FORMAT_AY64;
}
}                               // namespace schro

extern "C" {

// v210_bpp > 0: every plane goes to v210 from that depth; wide_bpp > 0: v216 / ARGB / AY64 by
// the plane's format from that depth; both 0: YUYV / UYVY / AYUV from u8
static int
pack_batch (SchroHipContext * ctx, const SchroHipPackPlane * planes, int nplanes, int v210_bpp, int wide_bpp = 0)
{
  SCHRO_HIP_REQUIRE (ctx && planes && nplanes > 0 && nplanes <= kMaxJobs, "pack_batch: bad arguments");
  (void) hipSetDevice (ctx->device);
  int gx, rows;
  pack_tile_geometry (&gx, &rows);
  std::vector < PackJob > jobs (nplanes);
  int tile_base = 0;
  for (int p = 0; p < nplanes; p++) {
    const SchroHipPackPlane & pl = planes[p];
    SCHRO_HIP_REQUIRE (pl.src[0] && pl.src[1] && pl.src[2] && pl.dst && pl.width > 0 && pl.height > 0
        && pl.src_width > 0 && pl.src_height > 0, "pack_batch: plane %d invalid", p);
    SCHRO_HIP_REQUIRE (v210_bpp || wide_bpp || pl.format == SCHRO_HIP_FORMAT_YUYV || pl.format == SCHRO_HIP_FORMAT_UYVY
        || pl.format == SCHRO_HIP_FORMAT_AYUV, "pack_batch: plane %d: format 0x%x is not YUYV / UYVY / AYUV",
        p, pl.format);
    if (wide_bpp) {
      SCHRO_HIP_REQUIRE (is_wide_format (pl.format), "pack_wide_batch: plane %d: format 0x%x is not v216 / ARGB / AY64",
          p, pl.format);
      // no chroma resampling after the depth conversion (schrovirtframe.c:1545-1575 knows u8 only)
      SCHRO_HIP_REQUIRE (pl.src_v_shift == 0 && pl.src_h_shift == (pl.format == SCHRO_HIP_FORMAT_v216 ? 1 : 0),
          "pack_wide_batch: plane %d: the source must be %s", p, pl.format == SCHRO_HIP_FORMAT_v216 ? "4:2:2" : "4:4:4");
    }
    // the reference resamples chroma of u8 frames only (schrovirtframe.c:1545-1575)
    SCHRO_HIP_REQUIRE (v210_bpp <= 1 || (pl.src_h_shift == 1 && pl.src_v_shift == 0),
        "pack_v210_batch: plane %d: s16 / s32 sources must be 4:2:2", p);
    SCHRO_HIP_REQUIRE ((pl.src_h_shift | pl.src_v_shift) >= 0 && pl.src_h_shift <= 1 && pl.src_v_shift <= 1
        && !(pl.src_v_shift && !pl.src_h_shift), "pack_batch: plane %d: chroma format not 4:4:4 / 4:2:2 / 4:2:0", p);
    // schroframe.c:931-941 crops both dimensions or extends both
    SCHRO_HIP_REQUIRE (!((pl.width < pl.src_width || pl.height < pl.src_height)
            && (pl.width > pl.src_width || pl.height > pl.src_height)),
        "pack_batch: plane %d: %dx%d from %dx%d mixes crop and extension", p, pl.width, pl.height,
        pl.src_width, pl.src_height);
    const int row_bytes = v210_bpp ? 16 * div_up (pl.width, 6)
        : wide_bpp ? (pl.format == SCHRO_HIP_FORMAT_v216 ? 8 * (pl.width / 2) : pl.format == SCHRO_HIP_FORMAT_ARGB
            ? 4 * pl.width : 8 * pl.width)
        : pl.format == SCHRO_HIP_FORMAT_AYUV ? 4 * pl.width : 4 * (pl.width / 2);
    SCHRO_HIP_REQUIRE (pl.dst_stride >= row_bytes, "pack_batch: plane %d stride too small", p);
    PackJob & j = jobs[p];
    for (int k = 0; k < 3; k++) {
      j.src[k] = pl.src[k];
      j.src_stride[k] = pl.src_stride[k];
    }
    j.dst = pl.dst;
    j.dst_stride = pl.dst_stride;
    j.sw = pl.src_width;
    j.sh = pl.src_height;
    j.hs = pl.src_h_shift;
    j.vs = pl.src_v_shift;
    j.w = pl.width;
    j.h = pl.height;
    j.format = v210_bpp ? SCHRO_HIP_FORMAT_v210 : pl.format;
    j.src_bpp = v210_bpp ? v210_bpp : (wide_bpp ? wide_bpp : 1);
    j.tiles_x = div_up (div_up (row_bytes, 16), gx);
    if (j.tiles_x == 0)
      j.tiles_x = 1;
    j.tile_base = tile_base;
    tile_base += j.tiles_x * div_up (pl.height, rows);
  }
  void *d_jobs;
  int r = push_args (ctx, jobs.data (), sizeof (PackJob) * nplanes, &d_jobs);
  if (r)
    return r;
  ProfileScope ps (ctx, SCHRO_HIP_KERNEL_CONVERT);
  return launch_pack (ctx->stream, (const PackJob *) d_jobs, nplanes, tile_base);
}

int
schro_hip_pack_u8_batch (SchroHipContext * ctx, const SchroHipPackPlane * planes, int nplanes)
{
  return pack_batch (ctx, planes, nplanes, 0);
}

int
schro_hip_pack_v210_batch (SchroHipContext * ctx, const SchroHipPackPlane * planes, int nplanes,
    int src_bpp)
{
  SCHRO_HIP_REQUIRE (src_bpp == 1 || src_bpp == 2 || src_bpp == 4, "pack_v210_batch: src_bpp must be 1, 2 or 4");
  return pack_batch (ctx, planes, nplanes, src_bpp);
}

int
schro_hip_pack_wide_batch (SchroHipContext * ctx, const SchroHipPackPlane * planes, int nplanes, int src_bpp)
{
  SCHRO_HIP_REQUIRE (src_bpp == 1 || src_bpp == 2 || src_bpp == 4, "pack_wide_batch: src_bpp must be 1, 2 or 4");
  return pack_batch (ctx, planes, nplanes, 0, src_bpp);
}

int
schro_hip_shift_right_batch (SchroHipContext * ctx, const SchroHipDcPlane * planes, int nplanes, int bytes_per_sample,
    int shift)
{
  SCHRO_HIP_REQUIRE (ctx && planes && nplanes > 0 && nplanes <= kMaxJobs, "shift_right_batch: bad arguments");
  SCHRO_HIP_REQUIRE (bytes_per_sample == 2 || bytes_per_sample == 4, "shift_right_batch: bytes_per_sample must be 2 or 4");
  SCHRO_HIP_REQUIRE (shift >= 0 && shift < 8 * bytes_per_sample, "shift_right_batch: shift %d", shift);
  if (shift == 0)
    return 0;
  (void) hipSetDevice (ctx->device);
  int tw, th;
  convert_tile_geometry (&tw, &th);
  std::vector < ConvertJob > jobs (nplanes);
  int tile_base = 0;
  for (int p = 0; p < nplanes; p++) {
    const SchroHipDcPlane & pl = planes[p];
    SCHRO_HIP_REQUIRE (pl.data && pl.width > 0 && pl.height > 0 && pl.stride >= pl.width * bytes_per_sample
        && pl.stride % bytes_per_sample == 0 && (uintptr_t) pl.data % bytes_per_sample == 0,
        "shift_right_batch: plane %d invalid", p);
    ConvertJob & j = jobs[p];
    j.src = pl.data;
    j.dst = (uint8_t *) pl.data;
    j.src_stride = j.dst_stride = pl.stride;
    j.w = pl.width;
    j.h = pl.height;
    j.tiles_x = div_up (pl.width, tw);
    j.tile_base = tile_base;
    tile_base += j.tiles_x * div_up (pl.height, th);
  }
  void *d_jobs;
  int r = push_args (ctx, jobs.data (), sizeof (ConvertJob) * nplanes, &d_jobs);
  if (r)
    return r;
  ProfileScope ps (ctx, SCHRO_HIP_KERNEL_CONVERT);
  return launch_shift_right (ctx->stream, (const ConvertJob *) d_jobs, nplanes, tile_base, bytes_per_sample, shift);
}

static size_t
upsampled_bytes (int width, int height, int ps, int *stride)
{
  if (width <= 0 || height <= 0)
    return 0;
  const size_t st = (size_t) hp_chunks (width, ps) * 512;
  if (stride)
    *stride = (int) st;
  return st * (size_t) div_up (height, kHpBandRows);
}

size_t
schro_hip_upsampled_bytes (int width, int height, int *stride)
{
  return upsampled_bytes (width, height, 0, stride);
}

size_t
schro_hip_upsampled_pair_bytes (int width, int height, int *stride)
{
  return upsampled_bytes (width, height, 1, stride);
}

static int
upsampled_download (SchroHipContext * ctx, void *const *host, int host_stride, const void *dev, int dev_stride, int width,
    int height, int ps)
{
  SCHRO_HIP_REQUIRE (ctx && host[0] && (!ps || host[1]) && dev && width > 0 && height > 0
      && dev_stride >= hp_chunks (width, ps) * 512 && dev_stride % 512 == 0 && host_stride >= 2 * width,
      "upsampled_download: bad arguments");
  (void) hipSetDevice (ctx->device);
  std::vector < uint8_t > raw ((size_t) dev_stride * (size_t) div_up (height, kHpBandRows));
  SCHRO_HIP_CHECK (hipMemcpyAsync (raw.data (), dev, raw.size (), hipMemcpyDeviceToHost, ctx->stream));
  SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
  for (int c = 0; c <= ps; c++)
    for (int y = 0; y < 2 * height; y++) {
      uint8_t *d = (uint8_t *) host[c] + (size_t) y * host_stride;
      for (int x = 0; x < 2 * width; x++)
        d[x] = raw[hp_offset (x, y, dev_stride, ps, c)];
    }
  return 0;
}

int
schro_hip_upsampled_download (SchroHipContext * ctx, void *host, int host_stride, const void *dev,
    int dev_stride, int width, int height)
{
  void *const hosts[2] = { host, nullptr };
  return upsampled_download (ctx, hosts, host_stride, dev, dev_stride, width, height, 0);
}

int
schro_hip_upsampled_pair_download (SchroHipContext * ctx, void *host_u, void *host_v, int host_stride, const void *dev,
    int dev_stride, int width, int height)
{
  void *const hosts[2] = { host_u, host_v };
  return upsampled_download (ctx, hosts, host_stride, dev, dev_stride, width, height, 1);
}

int
schro_hip_upsample_batch (SchroHipContext * ctx, const SchroHipUpsamplePlane * planes, int nplanes)
{
  SCHRO_HIP_REQUIRE (ctx && planes && nplanes > 0 && nplanes <= kMaxJobs,
      "upsample_batch: bad arguments");
  (void) hipSetDevice (ctx->device);
  int tw, th;
  upsample_tile_geometry (&tw, &th);
  std::vector < UpsampleJob > jobs (nplanes);
  int tile_base = 0;
  for (int p = 0; p < nplanes; p++) {
    const SchroHipUpsamplePlane & pl = planes[p];
    const int ps = pl.src_v ? 1 : 0;    // a (U, V) pair: samples of two bytes, tiles half as wide
    SCHRO_HIP_REQUIRE (pl.src && pl.dst && pl.width > 0 && pl.height > 0
        && pl.dst_stride >= hp_chunks (pl.width, ps) * 512 && pl.dst_stride % 512 == 0 && pl.src_stride >= pl.width
        && (!pl.src_v || pl.src_v_stride >= pl.width) && ((uintptr_t) pl.dst & 127) == 0,
        "upsample_batch: plane %d invalid (half-pel image: 128-byte aligned, stride from schro_hip_upsampled_bytes / _pair_bytes)", p);
    UpsampleJob & j = jobs[p];
    memset (&j, 0, sizeof (j));
    j.src = pl.src;
    j.dst = pl.dst;
    j.src_stride = pl.src_stride;
    j.dst_stride = pl.dst_stride;
    j.w = pl.width;
    j.h = pl.height;
    j.src_b = pl.src_v;
    j.src_b_stride = pl.src_v_stride;
    j.tiles_x = div_up (pl.width, tw >> ps);
    j.tile_base = tile_base;
    tile_base += j.tiles_x * div_up (pl.height, th);
  }
  void *d_jobs;
  int r = push_args (ctx, jobs.data (), sizeof (UpsampleJob) * nplanes, &d_jobs);
  if (r)
    return r;
  ProfileScope ps (ctx, SCHRO_HIP_KERNEL_UPSAMPLE);
  // scratch runs: SCHRO_HIP_UPSAMPLE_PERSIST = persistent workgroups per CU (launches whose planes are all of one form)
  static const int persist = SCHRO_ENV ("SCHRO_HIP_UPSAMPLE_PERSIST") ? atoi (SCHRO_ENV ("SCHRO_HIP_UPSAMPLE_PERSIST")) : 0;
  int grid = 0;
  if (persist > 0) {
    bool one_form = true;
    for (int p = 1; p < nplanes; p++)
      one_form = one_form && (!planes[p].src_v == !planes[0].src_v);
    if (one_form)
      grid = persist * ctx->cus / 8 * 8;
  }
  return launch_upsample (ctx->stream, (const UpsampleJob *) d_jobs, nplanes, tile_base, grid);
}

}                               // extern "C"
