// frame.cpp -- the C ABI of libschro_hip.so (include/schro_hip.h), third part: the frame layer -- the reference's
// own objects (SchroFrame / SchroParams / SchroMotion-shaped) and the stage calls a patched schrodecoder.c makes
// (schro_frame_inverse_iwt_transform_hip, schro_upsampled_hipframe_upsample, schro_motion_render_hip,
// schro_hipframe_convert ...), built on the plane layer (plane_*.cpp, iiwt_pack.cpp).

#include "schro_hip_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

using namespace schro;

static SchroHipContext *
frame_ctx (const SchroHipFrame * f)
{
  if (!f || !f->domain || !(f->domain->flags & SCHRO_MEMORY_DOMAIN_HIP))
    return nullptr;
  return f->domain->ctx;
}

extern "C" {

// ---- frame layer -----------------------------------------------------------------

// How a stage call ends.  The reference's scheduler expects a stage complete when its function returns
// (schroasync-pthread.c:320-328): the default -- the selected queue is waited for (that queue only: r03
// drained all four).  A host that pipelines pictures itself turns that off
// (schro_hip_context_set_stage_completion (ctx, 0)): the same calls then only enqueue on the selected
// queue, and marks / schro_hip_queue_synchronize order and end them.
static int
stage_done (SchroHipContext * ctx, int r)
{
  if (r || !ctx->stage_complete)
    return r;
  SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
  r = dc_gave_up (ctx);
  return r ? r : pred_overflow_poll (ctx);
}

int
schro_hip_context_set_stage_completion (SchroHipContext * ctx, int complete_on_return)
{
  SCHRO_HIP_REQUIRE (ctx, "set_stage_completion: no context");
  ctx->stage_complete = complete_on_return != 0;
  return 0;
}

static int
format_bpp (int format)
{
  switch (SCHRO_HIP_FORMAT_DEPTH (format)) {
    case SCHRO_HIP_FORMAT_DEPTH_U8:
      return 1;
    case SCHRO_HIP_FORMAT_DEPTH_S16:
      return 2;
    case SCHRO_HIP_FORMAT_DEPTH_S32:
      return 4;
  }
  return 0;
}

SchroHipFrame *
schro_hip_frame_new_and_alloc (SchroHipContext * ctx, int format, int width, int height,
    int upsampled)
{
  if (ctx && width > 0 && height > 0 && !upsampled && (format == SCHRO_HIP_FORMAT_YUYV
          || format == SCHRO_HIP_FORMAT_UYVY || format == SCHRO_HIP_FORMAT_AYUV || format == SCHRO_HIP_FORMAT_v210
          || is_wide_format (format))) {
    // packed output frame: one component (schroframe.c:81-99)
    SchroHipFrame *f = (SchroHipFrame *) calloc (1, sizeof (SchroHipFrame));
    f->refcount = 1;
    f->domain = ctx->domain;
    f->format = format;
    f->width = width;
    f->height = height;
    SchroHipFrameData *c = &f->components[0];
    c->format = format;
    c->width = width;
    c->height = height;
    const size_t row = format == SCHRO_HIP_FORMAT_AYUV || format == SCHRO_HIP_FORMAT_ARGB ? (size_t) width * 4
        : format == SCHRO_HIP_FORMAT_AY64 ? (size_t) width * 8
        : format == SCHRO_HIP_FORMAT_v216 ? (size_t) ((width + 1) & ~1) * 4
        : format == SCHRO_HIP_FORMAT_v210 ? (size_t) 16 * div_up (width, 6) : (size_t) ((width + 1) & ~1) * 2;
    c->stride = (int) round_up (row, 64);
    c->length = c->stride * height;
    c->data = schro_hip_domain_alloc (ctx, round_up ((size_t) c->length, 256));
    if (!c->data) {
      free (f);
      return nullptr;
    }
    f->regions[0] = c->data;
    return f;
  }
  int bpp = format_bpp (format);
  if (!ctx || !bpp || width <= 0 || height <= 0 || (format & 0x100) || (upsampled && bpp != 1)) {
    set_error (SCHRO_HIP_EINVAL, "frame_new_and_alloc: bad arguments");
    return nullptr;
  }
  SchroHipFrame *f = (SchroHipFrame *) calloc (1, sizeof (SchroHipFrame));
  f->refcount = 1;
  f->domain = ctx->domain;
  f->format = format;
  f->width = width;
  f->height = height;
  int h_shift = SCHRO_HIP_FORMAT_H_SHIFT (format), v_shift = SCHRO_HIP_FORMAT_V_SHIFT (format);
  // r04: the chroma of a horizontally subsampled upsampled frame is ONE pair image (is_upsampled == 2):
  // components[1] holds it, components[2] points at the same bytes (length 0: nothing of its own)
  const bool pair = upsampled && h_shift == 1;
  f->is_upsampled = pair ? 2 : upsampled ? 1 : 0;
  // chroma size rounds up, schroframe.c:95-96
  int cw = (width + (1 << h_shift) - 1) >> h_shift, ch = (height + (1 << v_shift) - 1) >> v_shift;
  size_t total = 0;
  for (int k = 0; k < 3; k++) {
    SchroHipFrameData *c = &f->components[k];
    c->format = format;
    c->width = k ? cw : width;
    c->height = k ? ch : height;
    c->h_shift = k ? h_shift : 0;
    c->v_shift = k ? v_shift : 0;
    c->stride = (int) round_up ((size_t) c->width * bpp, 64);
    c->length = c->stride * c->height;
    if (upsampled) {            // the four half-pel planes, tiled (include/schro_hip.h): stride = bytes per band of 4 rows
      int st = 0;
      c->length = (int) (pair && k ? schro_hip_upsampled_pair_bytes (c->width, c->height, &st)
          : schro_hip_upsampled_bytes (c->width, c->height, &st));
      c->stride = st;
      if (pair && k == 2)
        c->length = 0;
    }
    total += round_up ((size_t) c->length, 256);
  }
  void *base = schro_hip_domain_alloc (ctx, total);
  if (!base) {
    free (f);
    return nullptr;
  }
  f->regions[0] = base;
  size_t off = 0;
  for (int k = 0; k < 3; k++) {
    f->components[k].data = (char *) base + off;
    off += round_up ((size_t) f->components[k].length, 256);
  }
  if (pair)
    f->components[2].data = f->components[1].data;
  return f;
}

SchroHipFrame *
schro_hip_frame_ref (SchroHipFrame * frame)
{
  if (frame)
    frame->refcount++;
  return frame;
}

void
schro_hip_frame_unref (SchroHipFrame * frame)
{
  if (!frame)
    return;
  if (--frame->refcount > 0)
    return;
  if (frame_ctx (frame) && frame->regions[0])
    schro_hip_domain_free (frame_ctx (frame), frame->regions[0]);
  free (frame);
}

// all components of a frame, host <-> device or device -> device, on the selected queue; no wait
static int
copy_frame_async (SchroHipContext * ctx, SchroHipFrame * dest, const SchroHipFrame * src, hipMemcpyKind kind)
{
  if ((src->format & 0x100) || (dest->format & 0x100)) {
    SCHRO_HIP_REQUIRE (src->format == dest->format, "frame copy: packed format mismatch");
    const SchroHipFrameData *s = &src->components[0];
    SchroHipFrameData *d = &dest->components[0];
    int w = std::min (s->width, d->width), h = std::min (s->height, d->height);
    const int f = src->format;
    size_t row = f == SCHRO_HIP_FORMAT_AYUV || f == SCHRO_HIP_FORMAT_ARGB ? (size_t) w * 4
        : f == SCHRO_HIP_FORMAT_AY64 ? (size_t) w * 8 : f == SCHRO_HIP_FORMAT_v216 ? (size_t) (w / 2) * 8
        : f == SCHRO_HIP_FORMAT_v210 ? (size_t) 16 * div_up (w, 6) : (size_t) (w / 2) * 4;
    if (row && h > 0)
      return copy_2d_async (ctx, d->data, d->stride, s->data, s->stride, (int) row, h, kind);
    return 0;
  }
  int bpp = format_bpp (src->format);
  SCHRO_HIP_REQUIRE (bpp && format_bpp (dest->format) == bpp, "frame copy: depth mismatch");
  // (r04) Frames as the domains hand them out -- the components one behind the other in ONE block, the same sizes and
  // strides on both sides -- cross as one copy: three copies per 2160p frame cost the pipelined frame layer 0.05 ms per
  // picture.  (The row padding of such a frame belongs to it: copying it too is harmless.)
  {
    bool one = true;
    size_t total = 0;
    for (int k = 0; k < 3 && one; k++) {
      const SchroHipFrameData *s = &src->components[k];
      const SchroHipFrameData *d = &dest->components[k];
      one = s->data && d->data && s->width == d->width && s->height == d->height && s->width > 0 && s->height > 0
          && s->stride == d->stride && s->stride >= s->width * bpp && s->length == d->length
          && (size_t) s->length == (size_t) s->stride * s->height
          && (const uint8_t *) s->data == (const uint8_t *) src->components[0].data + total
          && (const uint8_t *) d->data == (const uint8_t *) dest->components[0].data + total;
      total += (size_t) s->length;
    }
    if (one && total < ((size_t) 1 << 31))
      return copy_2d_async (ctx, dest->components[0].data, (int) total, src->components[0].data, (int) total, (int) total, 1, kind);
  }
  for (int k = 0; k < 3; k++) {
    const SchroHipFrameData *s = &src->components[k];
    SchroHipFrameData *d = &dest->components[k];
    int w = std::min (s->width, d->width), h = std::min (s->height, d->height);
    if (w <= 0 || h <= 0)
      continue;
    int r = copy_2d_async (ctx, d->data, d->stride, s->data, s->stride, w * bpp, h, kind);
    if (r)
      return r;
  }
  return 0;
}

static int
copy_frame (SchroHipContext * ctx, SchroHipFrame * dest, const SchroHipFrame * src, hipMemcpyKind kind)
{
  int r = copy_frame_async (ctx, dest, src, kind);
  if (r)
    return r;
  SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
  return 0;
}

int
schro_frame_to_hip (SchroHipFrame * dest, SchroHipFrame * src)
{
  SCHRO_HIP_REQUIRE (dest && src && frame_ctx (dest) && !frame_ctx (src),
      "frame_to_hip: dest must be a device frame and src a host frame");
  return copy_frame (frame_ctx (dest), dest, src, hipMemcpyHostToDevice);
}

int
schro_hipframe_to_cpu (SchroHipFrame * dest, SchroHipFrame * src)
{
  SCHRO_HIP_REQUIRE (dest && src && frame_ctx (src) && !frame_ctx (dest),
      "hipframe_to_cpu: src must be a device frame and dest a host frame");
  return copy_frame (frame_ctx (src), dest, src, hipMemcpyDeviceToHost);
}

// r03 -- the asynchronous twins (TODO-CUDA:5-7 "make gpu stuff completely asynchronous"; the
// synchronous pattern they replace: schrogpuframe.c:480-609): enqueued on the context's selected
// queue -- SCHRO_HIP_QUEUE_H2D / _D2H by convention -- and not waited for.  The host frame should live in
// pinned memory (schro_memory_domain_new_hip_host): then picture k's copies run on the DMA engines
// beside picture k - 1's kernels; order them against the stages with marks and end with
// schro_hip_queue_synchronize.
int
schro_frame_to_hip_async (SchroHipFrame * dest, SchroHipFrame * src)
{
  SCHRO_HIP_REQUIRE (dest && src && frame_ctx (dest) && !frame_ctx (src),
      "frame_to_hip_async: dest must be a device frame and src a host frame");
  (void) hipSetDevice (frame_ctx (dest)->device);
  return copy_frame_async (frame_ctx (dest), dest, src, hipMemcpyHostToDevice);
}

int
schro_hipframe_to_cpu_async (SchroHipFrame * dest, SchroHipFrame * src)
{
  SCHRO_HIP_REQUIRE (dest && src && frame_ctx (src) && !frame_ctx (dest),
      "hipframe_to_cpu_async: src must be a device frame and dest a host frame");
  (void) hipSetDevice (frame_ctx (src)->device);
  return copy_frame_async (frame_ctx (src), dest, src, hipMemcpyDeviceToHost);
}

}                               // extern "C"

namespace schro {
// A copy of a device frame on another context's device (the scheduler moves a reference across two
// chains with it): same format, size and layout -- plain or upsampled --, one
// hipMemcpyPeerAsync per component on the destination context's host-to-device COPY queue, behind
// `wait_for` (the owner's event: all the work that writes `src`); `done` is recorded behind the copies
// and the destination's kernel queues wait for it.  Nothing is waited for on the host (r04).  (ROCm 7.2 holds the
// caller of an asynchronous copy from / to pinned HOST memory behind an unfired event; this PEER copy
// measured 22 us in the call with 7 ms of producer work outstanding -- two contexts on one device,
// profiles/r05_peer_copy_hip_trace.txt; between two real devices unmeasured.)
SchroHipFrame *
frame_copy_to_async (SchroHipContext * dst_ctx, SchroHipFrame * src, hipEvent_t wait_for, hipEvent_t done)
{
  SchroHipContext *src_ctx = frame_ctx (src);
  if (!dst_ctx || !src_ctx) {
    set_error (SCHRO_HIP_EINVAL, "frame_copy_to: needs a destination context and a device frame");
    return nullptr;
  }
  SchroHipFrame *dst = schro_hip_frame_new_and_alloc (dst_ctx, src->format, src->width, src->height, src->is_upsampled ? 1 : 0);
  if (!dst)
    return nullptr;
  (void) hipSetDevice (dst_ctx->device);
  hipStream_t q = dst_ctx->streams[SCHRO_HIP_QUEUE_H2D];
  const int ncomp = (src->format & 0x100) ? 1 : 3;
  bool ok = dst->is_upsampled == src->is_upsampled && (!wait_for || hipStreamWaitEvent (q, wait_for, 0) == hipSuccess);
  for (int k = 0; ok && k < ncomp; k++) {
    const SchroHipFrameData *s = &src->components[k];
    SchroHipFrameData *d = &dst->components[k];
    ok = d->length == s->length && d->stride == s->stride
        && (s->length == 0        // (the V component of a pair image: components[1] carries it)
        || hipMemcpyPeerAsync (d->data, dst_ctx->device, s->data, src_ctx->device, (size_t) s->length, q) == hipSuccess);
  }
  if (ok && done) {
    ok = hipEventRecord (done, q) == hipSuccess;
    for (int k = 0; ok && k < 2; k++)
      ok = hipStreamWaitEvent (dst_ctx->streams[k], done, 0) == hipSuccess;
  }
  if (!ok) {
    set_error (SCHRO_HIP_EDEVICE, "frame_copy_to: peer copy device %d -> %d failed", src_ctx->device, dst_ctx->device);
    (void) hipStreamSynchronize (q);
    schro_hip_frame_unref (dst);
    return nullptr;
  }
  dst->upsample_done = src->upsample_done;
  return dst;
}
}                               // namespace schro

extern "C" {

// the synchronous form: complete on return
SchroHipFrame *
schro_hip_frame_copy_to (SchroHipContext * dst_ctx, SchroHipFrame * src)
{
  SchroHipFrame *dst = schro::frame_copy_to_async (dst_ctx, src, nullptr, nullptr);
  if (dst && hipStreamSynchronize (dst_ctx->streams[SCHRO_HIP_QUEUE_H2D]) != hipSuccess) {
    set_error (SCHRO_HIP_EDEVICE, "frame_copy_to: the copy queue of device %d failed", dst_ctx->device);
    schro_hip_frame_unref (dst);
    return nullptr;
  }
  return dst;
}

// frame: the residual frame (s16 / s32, combine 0) or the u8 picture (combine 1: + prediction, 2: + 128)
static int
inverse_iwt_transform (SchroHipFrame * frame, SchroHipFrame * transform_frame, SchroHipParams * params, int combine,
    SchroHipFrame * prediction)
{
  SCHRO_HIP_REQUIRE (frame && transform_frame && params && frame_ctx (frame),
      "inverse_iwt_transform: bad arguments");
  SchroHipContext *ctx = frame_ctx (frame);
  int bpp = format_bpp (transform_frame->format);
  SCHRO_HIP_REQUIRE ((bpp == 2 || bpp == 4) && (combine ? format_bpp (frame->format) == 1 : format_bpp (frame->format) == bpp),
      "inverse_iwt_transform: the transform frame must be s16 or s32, the destination the same (the u8 picture in the combine form)");
  SCHRO_HIP_REQUIRE (combine != 1 || (prediction && prediction->domain == frame->domain && format_bpp (prediction->format) == 1),
      "inverse_iwt_transform: the combine form needs the u8 prediction frame of schro_motion_render_hip (add = FALSE) in the same domain");

  // host coefficients are staged on the device first (the H2D step of
  // schro_frame_inverse_iwt_transform_cuda, schrogpuframe.c:584-599)
  SchroHipFrame *staged = nullptr;
  SchroHipFrame *src = transform_frame;
  SCHRO_HIP_REQUIRE (ctx->stage_complete || frame_ctx (transform_frame),
      "inverse_iwt_transform: with stage completion off the transform frame must be on the device "
      "(schro_frame_to_hip_async on the copy queue)");
  if (!frame_ctx (transform_frame)) {
    staged = schro_hip_frame_new_and_alloc (ctx, transform_frame->format, transform_frame->width,
        transform_frame->height, 0);
    if (!staged)
      return SCHRO_HIP_ENOMEM;
    int r = schro_frame_to_hip (staged, transform_frame);
    if (r) {
      schro_hip_frame_unref (staged);
      return r;
    }
    src = staged;
  }
  SchroHipIwtPlane planes[3];
  memset (planes, 0, sizeof (planes));
  for (int k = 0; k < 3; k++) {
    planes[k].src = src->components[k].data;
    planes[k].src_stride = src->components[k].stride;
    planes[k].dst = frame->components[k].data;
    planes[k].dst_stride = frame->components[k].stride;
    planes[k].width = k ? params->iwt_chroma_width : params->iwt_luma_width;
    planes[k].height = k ? params->iwt_chroma_height : params->iwt_luma_height;
    bool fits = planes[k].width <= src->components[k].width && planes[k].height <= src->components[k].height;
    if (combine) {
      // the picture inside the transform's size (schrodecoder.c:1788-1790 converts with a crop; the render adds
      // over motion->width x height)
      planes[k].combine = combine;
      planes[k].out_width = std::min (frame->components[k].width, planes[k].width);
      planes[k].out_height = std::min (frame->components[k].height, planes[k].height);
      if (combine == 1) {
        planes[k].pred = (const uint8_t *) prediction->components[k].data;
        planes[k].pred_stride = prediction->components[k].stride;
        fits = fits && prediction->components[k].width >= planes[k].out_width && prediction->components[k].height >= planes[k].out_height;
      }
    } else {
      fits = fits && planes[k].width <= frame->components[k].width && planes[k].height <= frame->components[k].height;
    }
    if (!fits) {
      if (staged)
        schro_hip_frame_unref (staged);
      return set_error (SCHRO_HIP_EINVAL, "inverse_iwt_transform: component %d smaller than the iwt size", k);
    }
  }
  int r = schro_hip_iiwt_batch (ctx, planes, 3, params->transform_depth,
      params->wavelet_filter_index, bpp);
  r = stage_done (ctx, r);
  if (staged)
    schro_hip_frame_unref (staged);
  return r;
}

int
schro_frame_inverse_iwt_transform_hip (SchroHipFrame * frame,
    SchroHipFrame * transform_frame, SchroHipParams * params)
{
  return inverse_iwt_transform (frame, transform_frame, params, 0, nullptr);
}

// r04 -- x_wavelet_transform and the add of x_combine in one call (the structure of the reference's GPU paths:
// x_render_motion renders the prediction into mc_tmp_frame, x_combine adds, schrodecoder.c:1742-1760, :1908-1921):
// output = sat_u8 (inverse transform (transform_frame) + prediction), or + 128 where prediction is NULL (a picture
// without references, :1788-1790).  The residual picture never exists in memory.
int
schro_frame_inverse_iwt_transform_combine_hip (SchroHipFrame * output_frame, SchroHipFrame * transform_frame,
    SchroHipParams * params, SchroHipFrame * prediction)
{
  return inverse_iwt_transform (output_frame, transform_frame, params, prediction ? 1 : 2, prediction);
}

// r05 -- x_wavelet_transform and x_combine's schro_frame_convert into a v210 output picture in one call, for a picture
// without references (schrodecoder.c:1855-1886 + :2011-2052; the > 8-bit shift of :2013-2019 is 0 when the stream's depth
// is the output's): `packed` is a DEVICE frame of format v210, the transform frame a device s16 / s32 4:2:2 frame.  The
// pixel frame (picture->frame) is not written where the fused kernel applies (schro_hip_iiwt_pack_v210_batch).
int
schro_frame_inverse_iwt_transform_convert_hip (SchroHipFrame * packed, SchroHipFrame * transform_frame, SchroHipParams * params)
{
  SCHRO_HIP_REQUIRE (packed && transform_frame && params && frame_ctx (packed) && transform_frame->domain == packed->domain,
      "inverse_iwt_transform_convert: the packed frame and the transform frame must live in the same device domain");
  SCHRO_HIP_REQUIRE (packed->format == SCHRO_HIP_FORMAT_v210, "inverse_iwt_transform_convert: the destination must be a v210 frame "
      "(other formats: schro_frame_inverse_iwt_transform_hip + schro_hipframe_convert)");
  SchroHipContext *ctx = frame_ctx (packed);
  const int bpp = format_bpp (transform_frame->format);
  SCHRO_HIP_REQUIRE (bpp == 2 || bpp == 4, "inverse_iwt_transform_convert: the transform frame must be s16 or s32");
  SchroHipIwtPackPicture pic;
  for (int k = 0; k < 3; k++) {
    pic.src[k] = transform_frame->components[k].data;
    pic.src_stride[k] = transform_frame->components[k].stride;
  }
  pic.width = params->iwt_luma_width;
  pic.height = params->iwt_luma_height;
  pic.h_shift = SCHRO_HIP_FORMAT_H_SHIFT (transform_frame->format);
  pic.v_shift = SCHRO_HIP_FORMAT_V_SHIFT (transform_frame->format);
  SCHRO_HIP_REQUIRE (pic.width <= transform_frame->components[0].width && pic.height <= transform_frame->components[0].height
      && (params->iwt_chroma_width << pic.h_shift) == pic.width && (params->iwt_chroma_height << pic.v_shift) == pic.height,
      "inverse_iwt_transform_convert: the frame is smaller than params' transform size (or the chroma size is not the luma size shifted)");
  pic.dst = (uint8_t *) packed->components[0].data;
  pic.dst_stride = packed->components[0].stride;
  pic.out_width = packed->width;
  pic.out_height = packed->height;
  return stage_done (ctx, schro_hip_iiwt_pack_v210_batch (ctx, &pic, 1, params->transform_depth, params->wavelet_filter_index, bpp));
}

int
schro_hip_decode_lowdelay_transform_data (SchroHipFrame * transform_frame, const void *slices,
    size_t slices_bytes, const SchroHipLowDelayParams * params)
{
  SCHRO_HIP_REQUIRE (transform_frame && frame_ctx (transform_frame) && slices && params,
      "decode_lowdelay_transform_data: bad arguments");
  SchroHipContext *ctx = frame_ctx (transform_frame);
  const int bpp = format_bpp (transform_frame->format);
  SCHRO_HIP_REQUIRE (bpp == 2 || bpp == 4, "decode_lowdelay_transform_data: the frame must be s16 or s32");
  for (int k = 0; k < 3; k++)
    SCHRO_HIP_REQUIRE ((k ? params->iwt_chroma_width : params->iwt_luma_width) <= transform_frame->components[k].width
        && (k ? params->iwt_chroma_height : params->iwt_luma_height) <= transform_frame->components[k].height,
        "decode_lowdelay_transform_data: component %d smaller than the iwt size", k);
  // picture->lowdelay_buffer goes to the device as it is: compressed
  void *d_slices = schro_hip_domain_alloc (ctx, slices_bytes ? slices_bytes : 1);
  if (!d_slices)
    return SCHRO_HIP_ENOMEM;
  int r = 0;
  if (hipMemcpyAsync (d_slices, slices, slices_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
    r = set_error (SCHRO_HIP_EDEVICE, "decode_lowdelay_transform_data: copy of %zu bytes failed", slices_bytes);
  if (!r) {
    SchroHipLowDelayPicture pic;
    pic.slices = (const uint8_t *) d_slices;
    pic.slices_bytes = slices_bytes;
    for (int k = 0; k < 3; k++) {
      pic.comp[k] = transform_frame->components[k].data;
      pic.stride[k] = transform_frame->components[k].stride;
    }
    r = schro_hip_lowdelay_batch (ctx, &pic, 1, params, bpp);
  }
  const int rs = schro_hip_synchronize (ctx);   // the host buffer and d_slices are free again
  schro_hip_domain_free (ctx, d_slices);
  return r ? r : rs;
}

// r05 -- schro_decoder_decode_subband's data-parallel half on the device, behind the frame layer: the picture arrives
// as codeblock records + quantised values (include/schro_hip.h), leaves as the dense transform frame x_wavelet_transform
// reads.  Reference: schrodecoder.c:3525-3640 (the codeblock loop), :3311-3322 (zero codeblocks), :3060-3083 / :3400-3451
// (dequantisation), :3629-3636 + :3219-3277 (DC prediction of an intra picture's LL bands), :3280-3293 (codeblock counts).
int
schro_hipframe_dequantise (SchroHipFrame * transform_frame, const SchroHipQuantisedPicture * q, SchroHipParams * params)
{
  SCHRO_HIP_REQUIRE (transform_frame && frame_ctx (transform_frame) && q && params, "hipframe_dequantise: bad arguments "
      "(the transform frame must be a device frame)");
  SchroHipContext *ctx = frame_ctx (transform_frame);
  const int bpp = format_bpp (transform_frame->format);
  SCHRO_HIP_REQUIRE (bpp == 2 || bpp == 4, "hipframe_dequantise: the transform frame must be s16 or s32");
  (void) hipSetDevice (ctx->device);
  const int arith = params->is_noarith && bpp == 2 ? 1 : 0;
  const int intra = params->num_refs == 0;
  SchroHipDequantPlane planes[3];
  size_t stage_bytes = 0;
  for (int k = 0; k < 3; k++) {
    SCHRO_HIP_REQUIRE (q->codeblocks[k] && q->ncodeblocks[k] > 0 && (q->values[k] || q->values_bytes[k] == 0),
        "hipframe_dequantise: component %d has no codeblock records (or values_bytes without values)", k);
    stage_bytes += (q->values_bytes[k] + 255) & ~(size_t) 255;
    // r06: the records come, in the end, from the bitstream -- every codeblock lies inside its component
    // and its values inside the blob, or nothing is launched (two cheap loops over records the plan compare reads anyway)
    const SchroHipFrameData & comp = transform_frame->components[k];
    const long long comp_bytes = comp.length > 0 ? (long long) comp.length : (long long) comp.stride * comp.height;
    for (int n = 0; n < q->ncodeblocks[k]; n++) {
      const SchroHipCodeblock & cb = q->codeblocks[k][n];
      const long long row = (long long) cb.width * bpp;
      SCHRO_HIP_REQUIRE (cb.width >= 0 && cb.height >= 0, "hipframe_dequantise: component %d, codeblock %d has a negative size", k, n);
      if (cb.width == 0 || cb.height == 0)      // (an empty codeblock of a tiny sub-band: nothing is read or written)
        continue;
      SCHRO_HIP_REQUIRE (cb.dst_offset >= 0 && cb.dst_stride >= comp.stride && comp.stride > 0
          && cb.dst_stride % comp.stride == 0 && cb.dst_offset % bpp == 0
          && (long long) (cb.dst_offset % comp.stride) + row <= (long long) comp.stride
          && (long long) cb.dst_offset + (long long) (cb.height - 1) * cb.dst_stride + row <= comp_bytes,
          "hipframe_dequantise: component %d, codeblock %d (%d x %d at byte %d, pitch %d) does not lie inside the transform frame's "
          "component (%d x %d, stride %d)", k, n, cb.width, cb.height, cb.dst_offset, cb.dst_stride, comp.width, comp.height, comp.stride);
      SCHRO_HIP_REQUIRE (cb.src_offset < 0 || ((cb.src_bytes == 1 || cb.src_bytes == 2 || cb.src_bytes == 4)
              && (unsigned long long) cb.src_offset + (unsigned long long) cb.width * cb.height * cb.src_bytes <= (unsigned long long) q->values_bytes[k]),
          "hipframe_dequantise: component %d, codeblock %d: its values (%d bytes each, from byte %d) do not lie inside the %zu bytes of values",
          k, n, (int) cb.src_bytes, cb.src_offset, (size_t) q->values_bytes[k]);
    }
  }
  // host-side values: staged in a buffer of the selected queue (the queue's order keeps a later call's copy behind this
  // call's kernel); a host that pipelines uploads the picture's blob itself, on the copy queue, and passes device pointers
  char *stage = nullptr;
  if (!q->values_on_device && stage_bytes) {
    void *&buf = ctx->dq_stage_q[ctx->cur];
    size_t & size = ctx->dq_stage_size_q[ctx->cur];
    if (size < stage_bytes) {
      if (buf) {
        SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
        SCHRO_HIP_CHECK (hipFree (buf));
        buf = nullptr;
        size = 0;
      }
      SCHRO_HIP_CHECK (hipMalloc (&buf, stage_bytes + stage_bytes / 4));
      size = stage_bytes + stage_bytes / 4;
    }
    stage = (char *) buf;
  }
  size_t off = 0;
  for (int k = 0; k < 3; k++) {
    SchroHipDequantPlane & pl = planes[k];
    pl.dst = transform_frame->components[k].data;
    pl.codeblocks = q->codeblocks[k];
    pl.ncodeblocks = q->ncodeblocks[k];
    pl.is_intra = intra;
    if (q->values_on_device || !q->values_bytes[k]) {
      pl.values = q->values[k];
    } else {
      SCHRO_HIP_CHECK (hipMemcpyAsync (stage + off, q->values[k], q->values_bytes[k], hipMemcpyHostToDevice, ctx->stream));
      pl.values = stage + off;
      off += (q->values_bytes[k] + 255) & ~(size_t) 255;
    }
  }
  // the plan of this picture geometry: kept while pictures of the same geometry follow each other
  int r = SCHRO_HIP_EINVAL;
  if (ctx->frame_dq_plan)
    r = schro_hip_dequant_plan_matches (ctx->frame_dq_plan, planes, 3, bpp, arith) ? 0 : SCHRO_HIP_EINVAL;
  if (r) {
    if (ctx->frame_dq_plan)
      schro_hip_dequant_plan_free (ctx->frame_dq_plan);
    ctx->frame_dq_plan = schro_hip_dequant_plan_new (ctx, planes, 3, bpp, arith);
    if (!ctx->frame_dq_plan)
      return SCHRO_HIP_EINVAL;
  }
  r = schro_hip_dequant_plan_run (ctx->frame_dq_plan, planes, 3);
  if (!r && intra) {
    SchroHipDcPlane ll[3];
    for (int k = 0; k < 3; k++) {
      const SchroHipFrameData & c = transform_frame->components[k];
      const int w = k ? params->iwt_chroma_width : params->iwt_luma_width, h = k ? params->iwt_chroma_height : params->iwt_luma_height;
      ll[k].data = c.data;
      ll[k].stride = c.stride << params->transform_depth;        // schro_subband_get_frame_data, schroparams.c:319-352
      ll[k].width = w >> params->transform_depth;
      ll[k].height = h >> params->transform_depth;
    }
    r = schro_hip_dc_predict_batch (ctx, ll, 3, bpp);
  }
  return stage_done (ctx, r);
}

int
schro_upsampled_hipframe_upsample (SchroHipFrame * dest, SchroHipFrame * src)
{
  SCHRO_HIP_REQUIRE (dest && src && frame_ctx (dest) && src->domain == dest->domain
      && dest->is_upsampled && !src->is_upsampled && format_bpp (src->format) == 1,
      "upsampled_hipframe_upsample: bad arguments");
  if (dest->upsample_done)      // schroframe.c:2006-2009
    return 0;
  SchroHipUpsamplePlane planes[3];
  const bool pair = dest->is_upsampled == 2;    // chroma as one pair image
  for (int k = 0; k < 3; k++) {
    SCHRO_HIP_REQUIRE (dest->components[k].width == src->components[k].width
        && dest->components[k].height == src->components[k].height,
        "upsampled_hipframe_upsample: size mismatch");
    planes[k].src = (const uint8_t *) src->components[k].data;
    planes[k].src_stride = src->components[k].stride;
    planes[k].dst = (uint8_t *) dest->components[k].data;
    planes[k].dst_stride = dest->components[k].stride;
    planes[k].width = src->components[k].width;
    planes[k].height = src->components[k].height;
    planes[k].src_v = nullptr;
    planes[k].src_v_stride = 0;
  }
  if (pair) {
    planes[1].src_v = planes[2].src;
    planes[1].src_v_stride = planes[2].src_stride;
  }
  int r = stage_done (frame_ctx (dest), schro_hip_upsample_batch (frame_ctx (dest), planes, pair ? 2 : 3));
  if (!r)
    dest->upsample_done = 1;
  return r;
}

// schro_upsampled_gpuframe_upsample (SchroFrame *) (schrogpuframe.h:29): one argument, the upsampled
// frame, whose integer-pel source is the frame it keeps in virt_frame1
int
schro_upsampled_hipframe_upsample_inplace (SchroHipFrame * frame)
{
  SCHRO_HIP_REQUIRE (frame && frame->is_upsampled && frame->virt_frame1,
      "upsampled_hipframe_upsample_inplace: needs an upsampled frame with its source frame in virt_frame1");
  return schro_upsampled_hipframe_upsample (frame, frame->virt_frame1);
}

int
schro_motion_render_hip (SchroHipMotion * motion, SchroHipFrame * dest, SchroHipFrame * addframe, int add,
    SchroHipFrame * output_frame)
{
  // add == FALSE (r04): the prediction alone into `dest`, a u8 device frame -- schro_motion_render_cuda (motion,
  // mc_tmp_frame) (schrodecoder.c:1759) --, for schro_frame_inverse_iwt_transform_combine_hip to add.  Else `dest`
  // is the CPU path's s16 scratch frame and not used: the accumulator lives in LDS here.
  // r06: ... or an S16 device frame, which receives the prediction - 128 -- the literal contract of
  // schro_motion_render_cuda (motion, mc_tmp_frame) (schrocuda.h:13, schrodecoder.c:1742-1760) and of the CPU call's
  // add = FALSE (orc_rrshift6_sub_s16_2d's d2, schromotion8.c:896-899): schro_hipframe_add (frame, mc_tmp_frame) and
  // schro_hipframe_convert then finish the picture as schrodecoder.c:1908-1910, 2011 do.  Any weights, any DC values.
  bool s16_dest = false;
  if (!add) {
    SCHRO_HIP_REQUIRE (dest && frame_ctx (dest) && (format_bpp (dest->format) == 1 || format_bpp (dest->format) == 2) && !addframe,
        "motion_render: add = FALSE renders the prediction into `dest`, a u8 or s16 device frame (addframe NULL)");
    output_frame = dest;
    s16_dest = format_bpp (dest->format) == 2;
  }
  // addframe NULL: nothing to add -- a zero_residual picture has no frame (schrodecoder.c:1800, :1861,
  // :1904-1906: the GPU paths take mc_tmp_frame as the combined frame); the prediction alone is clamped
  SCHRO_HIP_REQUIRE (motion && motion->params && motion->src1 && motion->motion_vectors
      && output_frame && frame_ctx (output_frame) && (!addframe || addframe->domain == output_frame->domain),
      "motion_render: bad arguments");
  const SchroHipParams *p = motion->params;
  if (p->have_global_motion)    // schromotion.c:113-118 routes this to another renderer
    return set_error (SCHRO_HIP_EUNSUPPORTED, "motion_render: global motion is not supported");
  SchroHipContext *ctx = frame_ctx (output_frame);
  const int upsampled = p->mv_precision > 0;
  SCHRO_HIP_REQUIRE ((motion->src1->is_upsampled != 0) == upsampled
      && (!motion->src2 || motion->src2->is_upsampled == motion->src1->is_upsampled),
      "motion_render: references must be %s for mv_precision %d",
      upsampled ? "upsampled frames" : "plain frames", p->mv_precision);
  if (p->num_refs == 1)         // schromotion8.c:711-713
    SCHRO_HIP_REQUIRE (p->picture_weight_2 == 1, "motion_render: one reference needs picture_weight_2 == 1");

  // SchroMotionVector array -> device (schrogpumotion.c:68-120 did a repack; the kernel reads the
  // 20-byte records as they are).  r04: through the context's pinned table mirrors (push_big_table: one
  // memcpy into pinned memory + an asynchronous copy on the queue, buffers used in turn) -- no allocation,
  // no pageable-memory copy and no wait per call; a motion whose vectors are ALREADY on the device (a
  // host that uploaded them on the copy queue) is used where it is.
  (void) hipSetDevice (ctx->device);
  size_t mv_bytes = (size_t) 20 * p->x_num_blocks * p->y_num_blocks;
  void *d_mvs = nullptr;
  // r05 -- the combine form routes itself.  A prediction that cannot be a u8 plane (weights with a gain above 1; a DC
  // block whose value lies outside [-128, 127]: the reference's s16 block arithmetic carries it, schromotion8.c:542-568,
  // and wraps) is answered BEFORE anything is launched: SCHRO_HIP_ENEEDS_RESIDUAL, and the caller runs the picture in
  // the residual order (schrodecoder.c:1742-1760's stages the other way round: INTEGRATION 3).  Vectors that are already
  // on the device cannot be looked at here; the launch flags them (schro_hip_obmc_batch, prediction_only).
  if (!add && !s16_dest && (p->picture_weight_1 < 0 || p->picture_weight_2 < 0
          || p->picture_weight_1 + p->picture_weight_2 > (1 << p->picture_weight_bits)))
    return set_status (SCHRO_HIP_ENEEDS_RESIDUAL, "motion_render (add = FALSE): picture weights %d, %d / 2^%d can predict beyond 8 bits: "
        "this picture takes the residual order", p->picture_weight_1, p->picture_weight_2, p->picture_weight_bits);
  {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes (&attr, motion->motion_vectors) == hipSuccess && attr.type == hipMemoryTypeDevice) {
      d_mvs = motion->motion_vectors;
    } else {
      (void) hipGetLastError ();        // (an ordinary host pointer is "invalid value" to the query)
      if (!add && !s16_dest) {
        // SchroMotionVector (schromotion.h:53-75): pred_mode in the two low bits of the first word, the DC
        // values as three int16 at byte 12.  (Blocks on the picture's rim store their DC as a uint8_t and would
        // fit; they are refused with the rest -- the residual order is exact for every picture.)
        const unsigned char *mvb = (const unsigned char *) motion->motion_vectors;
        const size_t nmv = (size_t) p->x_num_blocks * p->y_num_blocks;
        for (size_t i = 0; i < nmv; i++) {
          const unsigned char *r = mvb + 20 * i;
          if (r[0] & 3)
            continue;
          int16_t dc[3];
          memcpy (dc, r + 12, 6);
          if ((unsigned) (dc[0] + 128) > 255u || (unsigned) (dc[1] + 128) > 255u || (unsigned) (dc[2] + 128) > 255u)
            return set_status (SCHRO_HIP_ENEEDS_RESIDUAL, "motion_render (add = FALSE): block %zu has a DC value outside [-128, 127] "
                "(%d, %d, %d): its prediction does not fit 8 bits, this picture takes the residual order", i, dc[0], dc[1], dc[2]);
        }
      }
      int r = push_big_table (ctx, motion->motion_vectors, mv_bytes, &d_mvs);
      if (r)
        return r;
    }
  }
  SchroHipObmcPlane planes[3];
  const int res_bpp = addframe ? format_bpp (addframe->format) : 2;
  for (int k = 0; k < 3; k++) {
    SchroHipObmcPlane & pl = planes[k];
    memset (&pl, 0, sizeof (pl));
    pl.mvs = d_mvs;
    pl.x_num_blocks = p->x_num_blocks;
    pl.y_num_blocks = p->y_num_blocks;
    pl.xblen_luma = p->xblen_luma;
    pl.yblen_luma = p->yblen_luma;
    pl.xbsep_luma = p->xbsep_luma;
    pl.ybsep_luma = p->ybsep_luma;
    pl.mv_precision = p->mv_precision;
    pl.picture_weight_bits = p->picture_weight_bits;
    pl.picture_weight_1 = p->picture_weight_1;
    pl.picture_weight_2 = p->picture_weight_2;
    pl.chroma_h_shift = SCHRO_HIP_FORMAT_H_SHIFT (output_frame->format);
    pl.chroma_v_shift = SCHRO_HIP_FORMAT_V_SHIFT (output_frame->format);
    pl.component = k;
    pl.ref1 = (const uint8_t *) motion->src1->components[k].data;
    pl.ref1_stride = motion->src1->components[k].stride;
    if (motion->src2) {
      pl.ref2 = (const uint8_t *) motion->src2->components[k].data;
      pl.ref2_stride = motion->src2->components[k].stride;
    }
    if (addframe) {
      pl.residual = addframe->components[k].data;
      pl.residual_stride = addframe->components[k].stride;
    }
    pl.residual_bpp = res_bpp;
    pl.out = (uint8_t *) output_frame->components[k].data;
    pl.out_stride = output_frame->components[k].stride;
    pl.width = output_frame->components[k].width;
    pl.height = output_frame->components[k].height;
    if (s16_dest) {
      // (the reference hands over a frame of the transform's padded size; the picture is what the references cover)
      pl.width = std::min (pl.width, motion->src1->components[k].width);
      pl.height = std::min (pl.height, motion->src1->components[k].height);
    }
    pl.ref_pair = k && motion->src1->is_upsampled == 2;
    pl.prediction_only = add ? 0 : s16_dest ? 2 : 1;
  }
  return stage_done (ctx, schro_hip_obmc_batch (ctx, planes, 3));
}

int
schro_hipframe_convert (SchroHipFrame * dest, SchroHipFrame * src)
{
  SCHRO_HIP_REQUIRE (dest && src && frame_ctx (dest) && src->domain == dest->domain,
      "hipframe_convert: both frames must live in the same device domain");
  SchroHipContext *ctx = frame_ctx (dest);
  if (dest->format & 0x100) {
    // copy-out into a packed frame (schroframe.c:878-899, 943-968)
    const bool wide = is_wide_format (dest->format), v210 = dest->format == SCHRO_HIP_FORMAT_v210;
    SCHRO_HIP_REQUIRE (!(src->format & 0x100) && format_bpp (src->format)
        && (wide || v210 || format_bpp (src->format) == 1),
        "hipframe_convert: YUYV / UYVY / AYUV take a planar u8 source (convert to u8 first)");
    SchroHipPackPlane pl;
    for (int k = 0; k < 3; k++) {
      pl.src[k] = (const uint8_t *) src->components[k].data;
      pl.src_stride[k] = src->components[k].stride;
    }
    pl.src_width = src->width;
    pl.src_height = src->height;
    pl.src_h_shift = SCHRO_HIP_FORMAT_H_SHIFT (src->format);
    pl.src_v_shift = SCHRO_HIP_FORMAT_V_SHIFT (src->format);
    pl.dst = (uint8_t *) dest->components[0].data;
    pl.dst_stride = dest->components[0].stride;
    pl.width = dest->width;
    pl.height = dest->height;
    pl.format = dest->format;
    int r = wide ? schro_hip_pack_wide_batch (ctx, &pl, 1, format_bpp (src->format))
        : v210 ? schro_hip_pack_v210_batch (ctx, &pl, 1, format_bpp (src->format)) : schro_hip_pack_u8_batch (ctx, &pl, 1);
    return stage_done (ctx, r);
  }
  int sb = format_bpp (src->format), db = format_bpp (dest->format);
  if (db == 1 && (sb == 2 || sb == 4)) {
    SchroHipConvertPlane planes[3];
    for (int k = 0; k < 3; k++) {
      planes[k].src = src->components[k].data;
      planes[k].src_stride = src->components[k].stride;
      planes[k].dst = (uint8_t *) dest->components[k].data;
      planes[k].dst_stride = dest->components[k].stride;
      planes[k].width = std::min (dest->components[k].width, src->components[k].width);
      planes[k].height = std::min (dest->components[k].height, src->components[k].height);
    }
    return stage_done (ctx, schro_hip_convert_u8_batch (ctx, planes, 3, sb));
  }
  if (db == sb) {
    (void) hipSetDevice (ctx->device);
    for (int k = 0; k < 3; k++) {
      int w = std::min (dest->components[k].width, src->components[k].width);
      int h = std::min (dest->components[k].height, src->components[k].height);
      SCHRO_HIP_CHECK (hipMemcpy2DAsync (dest->components[k].data, dest->components[k].stride,
              src->components[k].data, src->components[k].stride, (size_t) w * sb, h,
              hipMemcpyDeviceToDevice, ctx->stream));
    }
    return stage_done (ctx, 0);
  }
  return set_error (SCHRO_HIP_EUNSUPPORTED, "hipframe_convert: depth %d -> %d is not on the decode path",
      sb, db);
}

// schro_gpuframe_add (dest, src) (schrogpuframe.h:18, call site schrodecoder.c:1908-1910) = schro_frame_add
// (schroframe.c:1000-1029): dest (s16) += src (s16 | u8) over the components' common size
int
schro_hipframe_add (SchroHipFrame * dest, SchroHipFrame * src)
{
  SCHRO_HIP_REQUIRE (dest && src && frame_ctx (dest) && src->domain == dest->domain,
      "hipframe_add: both frames must live in the same device domain");
  SCHRO_HIP_REQUIRE (!(dest->format & 0x100) && !(src->format & 0x100) && format_bpp (dest->format) == 2
      && (format_bpp (src->format) == 1 || format_bpp (src->format) == 2)
      && SCHRO_HIP_FORMAT_H_SHIFT (dest->format) == SCHRO_HIP_FORMAT_H_SHIFT (src->format)
      && SCHRO_HIP_FORMAT_V_SHIFT (dest->format) == SCHRO_HIP_FORMAT_V_SHIFT (src->format),
      "hipframe_add: s16 += s16 | u8 of the same chroma format (add function unimplemented, schroframe.c:1027)");
  SchroHipConvertPlane planes[3];
  for (int k = 0; k < 3; k++) {
    planes[k].src = src->components[k].data;
    planes[k].src_stride = src->components[k].stride;
    planes[k].dst = (uint8_t *) dest->components[k].data;
    planes[k].dst_stride = dest->components[k].stride;
    planes[k].width = std::min (dest->components[k].width, src->components[k].width);
    planes[k].height = std::min (dest->components[k].height, src->components[k].height);
  }
  return stage_done (frame_ctx (dest), schro_hip_add_batch (frame_ctx (dest), planes, 3, format_bpp (src->format)));
}

int
schro_hipframe_shift_right (SchroHipFrame * frame, int shift)
{
  SCHRO_HIP_REQUIRE (frame && frame_ctx (frame) && !(frame->format & 0x100) && format_bpp (frame->format) > 1,
      "hipframe_shift_right: needs a device s16 / s32 frame");
  SchroHipDcPlane planes[3];
  for (int k = 0; k < 3; k++) {
    planes[k].data = frame->components[k].data;
    planes[k].stride = frame->components[k].stride;
    planes[k].width = frame->components[k].width;
    planes[k].height = frame->components[k].height;
  }
  return stage_done (frame_ctx (frame), schro_hip_shift_right_batch (frame_ctx (frame), planes, 3, format_bpp (frame->format), shift));
}

}                               // extern "C"
