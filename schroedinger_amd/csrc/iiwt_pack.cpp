// iiwt_pack.cpp -- r05: the inverse wavelet and the packed copy-out of an intra picture as ONE call (SURVEY 8f N2:
// "fusing it ... is required for 10-bit v210 output in config 5").
//
// The reference's chain for a picture without references whose output format is v210 (BASELINE config 5: VC-2 low delay,
// 10-bit 4:2:2): x_wavelet_transform writes the pixel frame (s32 for > 8-bit streams, schrodecoder.c:1855-1886), x_combine
// converts it into the application's picture (schro_frame_convert, :2011-2052 -> schrovirtframe.c:1438-1537 convert_s16_s32,
// :1823-1895 crop / edge extend, :943-991 pack_v210_s16).  On the device that was two launches with the 4-byte pixel frame
// written and read back between them: per 8K picture 265 MB out + 265 MB in around 88 MB of v210.  Where the transform is the
// three-level s32 Haar of a 4:2:2 picture (iiwt_haar.hip: an 8 x 8 block of samples depends on 64 coefficients of its own 8
// frame rows) the copy-out is the transform kernel's epilogue; every other picture takes the two passes, with the same bytes out.
#include "schro_hip_internal.h"

#include <cstring>
#include <vector>

using namespace schro;

extern "C" int
schro_hip_iiwt_pack_v210_batch (SchroHipContext * ctx, const SchroHipIwtPackPicture * pictures, int npictures, int depth, int filter,
    int bytes_per_sample)
{
  SCHRO_HIP_REQUIRE (ctx && pictures && npictures > 0 && npictures <= kMaxJobs, "iiwt_pack_v210_batch: bad arguments");
  SCHRO_HIP_REQUIRE (depth >= 1 && depth <= 6 && filter >= 0 && filter <= 6 && (bytes_per_sample == 2 || bytes_per_sample == 4),
      "iiwt_pack_v210_batch: depth %d, filter %d, %d bytes per sample", depth, filter, bytes_per_sample);
  (void) hipSetDevice (ctx->device);
  bool fused = depth == 3 && (filter == 3 || filter == 4) && bytes_per_sample == 4;
  std::vector < HaarPackJob > jobs ((size_t) npictures);
  int tile_base = 0;
  for (int p = 0; p < npictures; p++) {
    const SchroHipIwtPackPicture & pic = pictures[p];
    SCHRO_HIP_REQUIRE (pic.src[0] && pic.src[1] && pic.src[2] && pic.dst && pic.width > 0 && pic.height > 0
        && pic.out_width > 0 && pic.out_height > 0 && pic.out_width <= pic.width && pic.out_height <= pic.height
        && pic.dst_stride >= 16 * ((pic.out_width + 5) / 6), "iiwt_pack_v210_batch: picture %d: bad geometry", p);
    // (convert_4xx_422 of s16 / s32 frames is not on the reference's path either: schro_frame_convert finds no match, schroframe.c:869-979)
    SCHRO_HIP_REQUIRE (pic.h_shift == 1 && pic.v_shift == 0, "iiwt_pack_v210_batch: picture %d: v210 from s16 / s32 frames needs a 4:2:2 frame", p);
    HaarPackJob & j = jobs[(size_t) p];
    for (int c = 0; c < 3; c++) {
      j.src[c] = pic.src[c];
      j.src_stride[c] = pic.src_stride[c];
    }
    j.w = pic.width;
    j.h = pic.height;
    j.dst = pic.dst;
    j.dst_stride = pic.dst_stride;
    j.tiles_x = div_up (pic.width, iiwt_haar3_v210_strip_width ());
    j.tile_base = tile_base;
    tile_base += j.tiles_x * (pic.height / 8);
    fused = fused && pic.h_shift == 1 && pic.v_shift == 0 && pic.out_width == pic.width && pic.out_height == pic.height
        && iiwt_haar3_v210_ok (j);
  }
  if (fused) {
    void *d_jobs;
    int r = push_args (ctx, jobs.data (), sizeof (HaarPackJob) * jobs.size (), &d_jobs);
    if (r)
      return r;
    ProfileScope ps (ctx, SCHRO_HIP_KERNEL_IIWT_FINEST);
    return launch_iiwt_haar3_v210 (ctx->stream, (const HaarPackJob *) d_jobs, npictures, tile_base, filter);
  }
  // the two passes (the general form): transform into pixel planes, then the pack
  std::vector < SchroHipIwtPlane > planes ((size_t) 3 * npictures);
  std::vector < SchroHipPackPlane > packs ((size_t) npictures);
  // r06: the pixel planes live in the queue's own grow-only block -- nothing is allocated, freed or waited
  // for per call; a block that must grow waits for the queue once
  size_t need = 0;
  for (int p = 0; p < npictures; p++)
    for (int c = 0; c < 3; c++) {
      const SchroHipIwtPackPicture & pic = pictures[p];
      const int w = c ? pic.width >> pic.h_shift : pic.width, h = c ? pic.height >> pic.v_shift : pic.height;
      need += round_up (round_up ((size_t) w * bytes_per_sample, 64) * (size_t) h, 256);
    }
  const int q = ctx->cur;
  if (ctx->pack_tmp_size_q[q] < need) {
    if (ctx->pack_tmp_q[q]) {
      SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
      SCHRO_HIP_CHECK (hipFree (ctx->pack_tmp_q[q]));
      ctx->pack_tmp_q[q] = nullptr;
      ctx->pack_tmp_size_q[q] = 0;
    }
    SCHRO_HIP_CHECK (hipMalloc (&ctx->pack_tmp_q[q], need));
    ctx->pack_tmp_size_q[q] = need;
  }
  size_t at = 0;
  for (int p = 0; p < npictures; p++) {
    const SchroHipIwtPackPicture & pic = pictures[p];
    SchroHipPackPlane & pk = packs[(size_t) p];
    memset (&pk, 0, sizeof (pk));
    for (int c = 0; c < 3; c++) {
      const int w = c ? pic.width >> pic.h_shift : pic.width, h = c ? pic.height >> pic.v_shift : pic.height;
      const int stride = (int) round_up ((size_t) w * bytes_per_sample, 64);
      void *t = (char *) ctx->pack_tmp_q[q] + at;
      at += round_up ((size_t) stride * (size_t) h, 256);
      SchroHipIwtPlane & pl = planes[(size_t) 3 * p + c];
      memset (&pl, 0, sizeof (pl));
      pl.src = pic.src[c];
      pl.src_stride = pic.src_stride[c];
      pl.dst = t;
      pl.dst_stride = stride;
      pl.width = w;
      pl.height = h;
      pk.src[c] = (const uint8_t *) t;
      pk.src_stride[c] = stride;
    }
    pk.src_width = pic.out_width;       // (the picture inside the transform's padded size: crop, schrovirtframe.c:1823-1853)
    pk.src_height = pic.out_height;
    pk.src_h_shift = pic.h_shift;
    pk.src_v_shift = pic.v_shift;
    pk.dst = pic.dst;
    pk.dst_stride = pic.dst_stride;
    pk.width = pic.out_width;
    pk.height = pic.out_height;
    pk.format = SCHRO_HIP_FORMAT_v210;
  }
  int r = schro_hip_iiwt_batch (ctx, planes.data (), 3 * npictures, depth, filter, bytes_per_sample);
  if (!r)
    r = schro_hip_pack_v210_batch (ctx, packs.data (), npictures, bytes_per_sample);
  return r;
}
