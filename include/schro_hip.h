/*
 * schro_hip.h -- C ABI of libschro_hip.so, the MI355X (gfx950) execution
 * domain for the Dirac/VC-2 decode pixel path:
 *
 *     inverse lifting wavelet (7 filters, s16/s32)  ->  reference half-pel
 *     upsampling  ->  OBMC prediction + residual add + u8 clamp
 *
 * It occupies the slot the reference reserves for a GPU back end
 * (schroedinger/schrocuda.h:8-16, schrogpuframe.h:13-31, called from the
 * stage bodies schrodecoder.c:1697-2141), but produces the CPU path's bits:
 * every entry point is bit-exact to the reference's Orc/C implementation.
 *
 * Two layers, both plain C (pointers + sizes, no C++/torch types):
 *
 *   1. "plane" layer  -- batched kernel launches over device pointers.  This
 *      is what a host that already owns device memory binds (and what
 *      bench.py times).
 *   2. "frame" layer  -- SchroFrame-shaped structs and the stage-level calls
 *      a patched schrodecoder.c makes (INTEGRATION.md shows the patch).
 *
 * Error convention.  The reference has no return codes at this boundary:
 * device errors are SCHRO_ASSERT -> abort (schrogpuframe.c:249-253).  Here
 * every call returns 0 on success or a negative SCHRO_HIP_E* code and
 * records a message (schro_hip_last_error); schro_hip_set_abort_on_error(1)
 * restores the reference's abort behaviour.  All calls are synchronous with
 * respect to the context's stream only when they say so; the frame layer
 * returns after the work is complete, as the reference's stage scheduler
 * expects (schroasync-pthread.c:320-328).
 *
 * Threading: one SchroHipContext per exec-domain thread (the reference
 * creates exactly one such thread per GPU domain,
 * schroasync-pthread.c:362-390).  A context is not thread-safe.
 */
#ifndef SCHRO_HIP_H
#define SCHRO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCHRO_HIP_OK 0
#define SCHRO_HIP_EINVAL (-1)   /* bad argument (what SCHRO_ASSERT would trap) */
#define SCHRO_HIP_EDEVICE (-2)  /* HIP runtime error */
#define SCHRO_HIP_ENOMEM (-3)
#define SCHRO_HIP_EUNSUPPORTED (-4)
#define SCHRO_HIP_ESKIPPED (-5) /* scheduler: the picture did not run, one of its references had failed */
/* r05 -- not an error of the stream but a ROUTING answer of the combine form (add = FALSE / prediction_only): this
 * picture's prediction does not fit the u8 plane it would be written to (a DC value outside [-128, 127], or picture
 * weights with a gain above 1 -- the reference's 16-bit block arithmetic wraps there, schromotion8.c:542-657), so
 * the picture takes the residual order instead (inverse transform into the residual frame, then
 * schro_motion_render_hip (..., add = TRUE, ...): INTEGRATION 3).  schro_motion_render_hip (add = FALSE) answers
 * BEFORE it launches anything when the vectors are in host memory (it scans them); with device-resident vectors the
 * launch raises a flag that the next SYNCHRONISING call of the context (a stage call with stage completion on,
 * schro_hip_synchronize, schro_hip_queue_synchronize) reports once with this code, naming the prediction_only
 * batches (schro_hip_obmc_prediction_epoch numbers them); r06: a call that only enqueues is never refused for it, and
 * schro_hip_obmc_overflowed returns every such batch's number, so a host that pipelines pictures knows exactly
 * which ones to repeat.  schro_hip_set_abort_on_error (1) never turns this code into an abort. */
#define SCHRO_HIP_ENEEDS_RESIDUAL (-6)

/* schroedinger/schrodomain.h:30-36 -- ids for the new domain */
#define SCHRO_EXEC_DOMAIN_HIP 0x0004
#define SCHRO_MEMORY_DOMAIN_HIP 0x0008

/* SchroFrameFormat bits, schroedinger/schroframe.h:16-54 */
#define SCHRO_HIP_FORMAT_DEPTH(f) ((f) & 0xc)
#define SCHRO_HIP_FORMAT_DEPTH_U8 0x00
#define SCHRO_HIP_FORMAT_DEPTH_S16 0x04
#define SCHRO_HIP_FORMAT_DEPTH_S32 0x08
#define SCHRO_HIP_FORMAT_H_SHIFT(f) ((f) & 0x1)
#define SCHRO_HIP_FORMAT_V_SHIFT(f) (((f) >> 1) & 0x1)

typedef struct SchroHipContext SchroHipContext;

/* ---- context / memory domain -------------------------------------------- */

/* replaces schro_cuda_init (schrocuda.h:8) + schro_memory_domain_new_cuda
 * (schrocuda.h:9): binds `device`, creates the stream and the size-keyed
 * device allocation cache (schrodomain.c:58-137 semantics: freed blocks are
 * kept and handed back to the next request of the same size). */
SchroHipContext *schro_hip_context_new (int device);
void schro_hip_context_free (SchroHipContext * ctx);
int schro_hip_device_count (void);
const char *schro_hip_last_error (void);
void schro_hip_set_abort_on_error (int enable);
/* schro_cuda_init (schrocuda.h:8) twin: looks at the devices (SCHRO_HIP_DEBUG prints them) */
void schro_hip_init (void);
/* The calling thread becomes the exec-domain thread of `ctx` (NULL: of none): the alloc / free table
 * of a SchroMemoryDomain carries no domain argument (schrodomain.h:18-22), so it serves the domain the
 * calling THREAD is bound to -- and fails loudly on a thread bound to none.  schro_hip_context_new
 * binds the creating thread; the scheduler's threads bind theirs. */
void schro_hip_thread_bind (SchroHipContext * ctx);
SchroHipContext *schro_hip_thread_bound (void);        /* the calling thread's domain, NULL if none */

/* SchroMemoryDomain.alloc / .free (schrodomain.h:18-22) */
void *schro_hip_domain_alloc (SchroHipContext * ctx, size_t size);
int schro_hip_domain_free (SchroHipContext * ctx, void *ptr);
/* bytes currently held by the domain (in use + cached) */
size_t schro_hip_domain_bytes (SchroHipContext * ctx);

/* host<->device copies on the context stream; *_sync wait for completion.
 * 2-D copies move `height` rows of `row_bytes`. */
int schro_hip_upload_2d (SchroHipContext * ctx, void *dst, int dst_stride,
    const void *src, int src_stride, int row_bytes, int height);
int schro_hip_download_2d (SchroHipContext * ctx, void *dst, int dst_stride,
    const void *src, int src_stride, int row_bytes, int height);
int schro_hip_memset (SchroHipContext * ctx, void *dst, int value,
    size_t bytes);
/* r03 -- asynchronous transfers (TODO-CUDA:5-7; the synchronous pattern of schrogpuframe.c:480-609).
 * Pinned host memory is what the DMA engines copy from / to at full rate without the host thread:
 * schro_hip_host_alloc for blobs, schro_memory_domain_new_hip_host () for the reference's frames
 * (schro_frame_new_and_alloc (domain, ...) then hands out pinned host frames: ordinary host memory to
 * every CPU stage).  The _async copies are enqueued on the SELECTED queue and not waited for; by
 * convention uploads go to SCHRO_HIP_QUEUE_H2D and downloads to SCHRO_HIP_QUEUE_D2H, so that picture
 * k + 1's coefficients go up and picture k - 1's pixels come down beside picture k's kernels on
 * queues 0 / 1; marks order them (upload -> mark -> the wavelet's queue waits for it; OBMC -> mark ->
 * the download queue waits for it); schro_hip_queue_synchronize waits for one queue.
 * r04 -- ORDER OF THE CALLS.  On ROCm 7.2 an asynchronous copy enqueued on a queue that waits for an event which
 * has not fired yet returns only when it has (kernels behind such a wait return at once).  A host that wants its
 * thread back enqueues a copy when everything the copy waits for has already happened: hand over picture
 * k - 2 (schro_hip_queue_mark_synchronize on ITS download mark) before enqueueing anything of picture k, and enqueue
 * the download of picture k - 1 once its kernels' mark has fired (INTEGRATION.md 3a has the loop; DESIGN.md 5 the
 * measurements: 0.2 instead of 1.4 ms of host time per step of 8 x 2160p). */
void *schro_hip_host_alloc (size_t size);
void schro_hip_host_free (void *ptr);
int schro_hip_upload_2d_async (SchroHipContext * ctx, void *dst, int dst_stride,
    const void *src, int src_stride, int row_bytes, int height);
int schro_hip_download_2d_async (SchroHipContext * ctx, void *dst, int dst_stride,
    const void *src, int src_stride, int row_bytes, int height);
int schro_hip_queue_synchronize (SchroHipContext * ctx, int queue);
/* Restricts a queue's kernels to the compute units whose bits are set in mask[0 .. words) (the queue's
 * pending work is waited for first; hipExtStreamCreateWithCUMask): two queues with disjoint masks run
 * side by side without sharing a CU's registers and LDS.
 * The queue becomes a NEW stream, and one with default flags (the runtime offers no non-blocking form of
 * this call): unlike the context's other queues it synchronises implicitly with the process's NULL stream,
 * so a synchronous hipMemcpy / hipMemset anywhere in the process serialises against it.  This library's
 * per-picture calls issue every copy, memset and kernel on an explicit queue; its set-up calls
 * (schro_hip_dequant_plan_new, a scratch buffer that grows) use synchronous ones.  If the runtime refuses the mask the
 * old queue stays in place (drained) and the call returns SCHRO_HIP_EDEVICE.  Measured slower than whole
 * batches per queue for this path: a tool for experiments, not part of the decode loop. */
int schro_hip_queue_set_cu_mask (SchroHipContext * ctx, int queue, const uint32_t * mask, int words);
/* waits for everything enqueued on both queues */
int schro_hip_synchronize (SchroHipContext * ctx);
/* the selected queue's hipStream_t, for hosts that enqueue their own work or events */
void *schro_hip_stream (SchroHipContext * ctx);

/* In-order queues per context (r03: four; 0 and 1 carry the kernels, 2 and 3 the copies).  The reference's scheduler runs the stages of different
 * pictures on several worker threads at once (schroasync-pthread.c:320-390,
 * schro_decoder_async_schedule, schrodecoder.c:1546-1682); on this domain the same
 * concurrency is two queues: the inverse wavelet is HBM-bound, OBMC is issue-bound, so
 * picture batch k's schro_hip_obmc_batch on one queue and batch k+1's schro_hip_iiwt_batch /
 * schro_hip_upsample_batch on the other run side by side.  Every call of this header goes to
 * the selected queue (0 after schro_hip_context_new); schro_hip_queue_wait makes the work
 * enqueued LATER on `waiter` start after everything enqueued SO FAR on `signaller` (the
 * render_ok / wavelet-done dependencies of schrodecoder.c:1589-1660). */
#define SCHRO_HIP_QUEUES 4
#define SCHRO_HIP_QUEUE_H2D 2      /* by convention: host-to-device copies */
#define SCHRO_HIP_QUEUE_D2H 3      /* device-to-host copies */
int schro_hip_context_select_queue (SchroHipContext * ctx, int queue);
int schro_hip_context_queue (SchroHipContext * ctx);
int schro_hip_queue_wait (SchroHipContext * ctx, int waiter, int signaller);
/* finer than "everything so far": schro_hip_queue_mark records mark m (0 .. 15) behind the work
 * enqueued so far on the SELECTED queue; schro_hip_queue_wait_mark makes the selected queue's
 * later work wait for the most recent recording of m (no-op if m was never recorded).  E.g.
 * batch k + 2's wavelet may overwrite batch k's residual frames once batch k's OBMC is done. */
#define SCHRO_HIP_MARKS 16
int schro_hip_queue_mark (SchroHipContext * ctx, int mark);
int schro_hip_queue_wait_mark (SchroHipContext * ctx, int mark);
/* the HOST waits for the latest recording of `mark` (never recorded: returns at once) */
int schro_hip_queue_mark_synchronize (SchroHipContext * ctx, int mark);

/* HIP-event timing of everything enqueued between begin and end on the
 * context stream; schro_hip_timer_end synchronises and returns milliseconds
 * (< 0 on error). */
int schro_hip_timer_begin (SchroHipContext * ctx);
float schro_hip_timer_end (SchroHipContext * ctx);

/* Per-kernel HIP-event profiling.  When enabled, every kernel launch of the
 * plane layer carries a start / stop event pair of its own (hipExtLaunchKernelGGL:
 * the kernel's begin and end as the dispatch records them, nothing extra on the
 * queue); schro_hip_profile_read synchronises and returns the summed elapsed time
 * and the number of launches of one kernel class since the last reset.  bench.py
 * uses this for the roofline figure of the dominant kernel. */
#define SCHRO_HIP_KERNEL_IIWT_FINEST 0  /* level-0 launch of the inverse wavelet */
#define SCHRO_HIP_KERNEL_IIWT_COARSE 1  /* levels >= 1 */
#define SCHRO_HIP_KERNEL_UPSAMPLE 2
#define SCHRO_HIP_KERNEL_OBMC 3
#define SCHRO_HIP_KERNEL_CONVERT 4
#define SCHRO_HIP_KERNEL_SLICES 5
#define SCHRO_HIP_KERNEL_DC_PREDICT 6
#define SCHRO_HIP_KERNEL_DEQUANT 7
#define SCHRO_HIP_KERNEL_CLASSES 8
int schro_hip_profile_enable (SchroHipContext * ctx, int enable);
int schro_hip_profile_reset (SchroHipContext * ctx);
int schro_hip_profile_read (SchroHipContext * ctx, int kernel_class,
    double *total_ms, int *launches);

/* ---- plane layer: batched launches --------------------------------------- */

/* One component of one picture for the inverse wavelet.
 * Replaces the level loop schro_decoder_inverse_iwt_transform
 * (schrodecoder.c:1809-1853) / schro_gpuframe_inverse_iwt_transform
 * (schrogpuframe.c:399-478).  `src` holds the coefficients in the reference's
 * in-place sub-band layout (schroparams.c:319-352: level view
 * {w>>l, h>>l, stride<<l}; even rows = [LL|HL], odd rows = [LH|HH]);
 * `dst` receives the iwt_width x iwt_height result.  src and dst must not
 * overlap (the transform is tiled, not in place); src is left untouched. */
typedef struct {
  const void *src;
  int src_stride;               /* bytes */
  void *dst;
  int dst_stride;               /* bytes */
  int width;                    /* iwt_{luma,chroma}_width  */
  int height;                   /* iwt_{luma,chroma}_height */
  /* r04 -- the combine form: the transform's last step writes the PICTURE instead of the residual frame.
   * combine 0: dst is the s16 / s32 residual plane (width x height), as above.
   * combine 1: dst is a u8 plane, out_width x out_height (the picture inside the iwt-padded transform):
   *            dst = sat_u8 (residual + pred) with the reference's 16-bit wrapping add -- the last steps of
   *            schro_motion_render (..., add = TRUE, output_frame) (orc_rrshift6_add_s16_2d / _s32_2d,
   *            schromotion8.c:852-876) --, pred (u8, out_width x out_height) = the prediction
   *            schro_hip_obmc_batch writes with prediction_only = 1.
   * combine 2: dst = sat_u8 (residual + 128): a picture without references (schro_frame_convert,
   *            orc_offsetconvert_u8_s16 / _s32, schrodecoder.c:1788-1790).
   * The residual then never exists in memory (s16, register-form filters, 8-byte aligned planes; other
   * cases go through a residual plane in the context's scratch): a step of 8 x 2160p moves 2 x 199 MB less. */
  const uint8_t *pred;
  int pred_stride;
  int out_width, out_height;
  int combine;
  /* r04 -- a picture's transform in TWO calls.  ll != NULL: the LL band of this call's COARSEST level (level
   * depth - 1: (width >> depth) x (height >> depth) samples of the plane's type) is read from this compact plane
   * instead of from the frame's LL quadrant.  The levels above 0 do not depend on the picture's prediction, so a
   * host runs them early and beside other work -- call 1: src = the frame, src_stride = 2 x the frame's stride,
   * width / 2 x height / 2 (the level-1 view of the in-place layout, schroparams.c:319-352), depth - 1 levels,
   * dst = an LL plane of the host's; call 2, once the prediction exists: the frame itself, depth 1, ll = that
   * plane, combine as above.  The two calls together are the one call, bit for bit. */
  const void *ll;
  int ll_stride;                /* bytes */
  int reserved;
} SchroHipIwtPlane;

/* depth levels, Dirac filter index 0..6 (schrobitstream.h:124-132),
 * bpp 2 (s16) or 4 (s32).  width/height must be multiples of 1<<depth.
 * Enqueues `depth` launches (coarse to fine), each covering all planes. */
int schro_hip_iiwt_batch (SchroHipContext * ctx,
    const SchroHipIwtPlane * planes, int nplanes, int depth, int filter,
    int bpp);

/* intra pictures: dst_u8 = sat_u8 (src + 128), cropped to width x height;
 * replaces schro_frame_convert (ref_output_frame, frame)
 * (schrodecoder.c:1788-1790) / schro_gpuframe_convert. */
typedef struct {
  const void *src;
  int src_stride;
  uint8_t *dst;
  int dst_stride;
  int width;
  int height;
} SchroHipConvertPlane;

int schro_hip_convert_u8_batch (SchroHipContext * ctx,
    const SchroHipConvertPlane * planes, int nplanes, int bpp);

/* r06 -- dst (int16) += src (int16 when src_bytes_per_sample is 2, uint8 zero-extended when 1), 16-bit wrapping add
 * over width x height: schro_frame_add's two cases on planes (schroframe.c:1082-1135: orc_add_s16_2d,
 * orc_add_s16_u8_2d) = schro_gpuframe_add's (schrogpuframe.c:257-306).  `dst` of the plane is the int16 plane. */
int schro_hip_add_batch (SchroHipContext * ctx, const SchroHipConvertPlane * planes, int nplanes,
    int src_bytes_per_sample);

/* Copy-out of a decoded u8 picture into a packed output frame; replaces the
 * packed-destination case of schro_frame_convert (&output_picture, ref_output_frame)
 * in x_combine (schrodecoder.c:2011, 2052 -> schroframe.c:869-979): chroma
 * resampled by sample repetition / decimation to the packed format's chroma
 * (convert_4xx_4yy, schrovirtframe.c:1438-1537), cropped or edge-extended to
 * width x height (:1823-1895), packed (pack_yuyv / _uyvy / _ayuv :943-991,
 * 1230-1247).  On the device this halves the bytes of the device-to-host copy
 * of an 8-bit picture for a host that asked for a packed format.
 * YUYV / UYVY write width / 2 four-byte groups per row, AYUV width. */
#define SCHRO_HIP_FORMAT_YUYV 0x100     /* SCHRO_FRAME_FORMAT_YUYV, schroframe.h:36-38 */
#define SCHRO_HIP_FORMAT_UYVY 0x101
#define SCHRO_HIP_FORMAT_AYUV 0x102
#define SCHRO_HIP_FORMAT_ARGB 0x103     /* from s16 4:4:4 YCoCg-R planes */
#define SCHRO_HIP_FORMAT_v216 0x105     /* from s16 4:2:2 */
#define SCHRO_HIP_FORMAT_v210 0x106     /* 10 bit 4:2:2, six pixels in four words */
#define SCHRO_HIP_FORMAT_AY64 0x107     /* 16-bit A, Y, U, V from s32 4:4:4 */

typedef struct {
  const uint8_t *src[3];        /* Y, U, V planes (device) */
  int src_stride[3];
  int src_width, src_height;    /* luma size of the source picture */
  int src_h_shift, src_v_shift; /* its chroma subsampling (0/1) */
  uint8_t *dst;                 /* packed rows (device) */
  int dst_stride;
  int width, height;            /* size of the packed picture; not wider AND shorter (or
                                 * narrower and taller) than the source: the reference
                                 * crops both dimensions or extends both */
  int format;                   /* SCHRO_HIP_FORMAT_YUYV / _UYVY / _AYUV */
} SchroHipPackPlane;

int schro_hip_pack_u8_batch (SchroHipContext * ctx,
    const SchroHipPackPlane * planes, int nplanes);

/* The same copy-out into v210 (schroframe.c:889-891, 960-962): rows of
 * ceil (width / 6) 16-byte groups, samples beyond `width` in the last group 0.
 * src_bpp 1: u8 planes of any chroma format, 10-bit value (x << 2) | (x >> 6)
 * (pack_v210, schrovirtframe.c:1129-1210).  src_bpp 2 / 4: s16 / s32 planes --
 * the frame of a > 8-bit stream (schrodecoder.c:350-352), e.g. the 10-bit 4:2:2
 * 8K configuration -- which must be 4:2:2 as in the reference (its chroma
 * resampler only knows u8, schrovirtframe.c:1545-1575); s32 is truncated to
 * 16 bits (orc_convert_s16_s32), value clamp (x + 512, 0, 1023) (pack_v210_s16,
 * :1044-1127).  `format` of the planes is ignored; `src` / `src_stride` are in
 * bytes of the sample type. */
int schro_hip_pack_v210_batch (SchroHipContext * ctx,
    const SchroHipPackPlane * planes, int nplanes, int src_bpp);

/* r05 -- the inverse wavelet of an intra picture AND its v210 copy-out as one call (the reference's chain:
 * x_wavelet_transform, schrodecoder.c:1855-1886, then schro_frame_convert (output_picture, frame) in x_combine, :2011-2052 ->
 * schrovirtframe.c:1438-1537, :943-991).  dst receives exactly the bytes of schro_hip_iiwt_batch into a pixel frame followed by
 * schro_hip_pack_v210_batch from it.  Where the transform is the three-level s32 Haar (filters 3, 4) of a 4:2:2 picture whose
 * size is a multiple of 192 x 8 (v210 groups of 6 pixels x whole strips of the kernel; BASELINE config 5: 7680 x 4320) the copy-out is the transform kernel's epilogue and the pixel
 * frame never exists (per 8K picture 353 MB of memory traffic instead of 883 MB); every other case runs the two passes. */
typedef struct {
  const void *src[3];           /* the coefficient planes Y, U, V (device, s16 or s32), in-place sub-band layout */
  int src_stride[3];
  int width, height;            /* luma transform size (a multiple of 2^depth); chroma: >> h_shift, >> v_shift */
  int h_shift, v_shift;         /* 1, 0: v210 is a 4:2:2 format and s16 / s32 frames are not resampled (as schro_hip_pack_v210_batch) */
  uint8_t *dst;                 /* v210: 16 bytes per 6 pixels, dst_stride bytes per row */
  int dst_stride;
  int out_width, out_height;    /* the picture inside the transform's size */
} SchroHipIwtPackPicture;
int schro_hip_iiwt_pack_v210_batch (SchroHipContext * ctx, const SchroHipIwtPackPicture * pictures, int npictures, int depth,
    int filter, int bytes_per_sample);

/* The remaining packed destinations of schro_frame_convert (schroframe.c:886-895, 957-968),
 * chosen by `format` of each plane:
 *   SCHRO_HIP_FORMAT_v216  from a 4:2:2 source brought to s16; pack_v216
 *                          (schrovirtframe.c:1007-1028) reads the s16 lines through byte
 *                          pointers and this does exactly the same: 8 bytes per pixel pair,
 *                          rows of width / 2 pairs;
 *   SCHRO_HIP_FORMAT_ARGB  from a 4:4:4 source brought to s16, YCoCg-R -> A,R,G,B bytes
 *                          (pack_argb :1265-1287), 4 bytes per pixel;
 *   SCHRO_HIP_FORMAT_AY64  from a 4:4:4 source brought to s32, 16-bit A,Y,U,V words
 *                          clamp (x + 0x8000, 0, 0xffff) (pack_ayuv64 :1302-1322), 8 bytes per pixel.
 * src_bpp 1 / 2 / 4 with the reference's depth conversions (u8: x - 128; s32 -> s16:
 * truncation; s16 -> s32: sign extension, schrovirtframe.c:1742-1817); the source's chroma
 * format must already be the destination's (the reference resamples u8 frames only). */
int schro_hip_pack_wide_batch (SchroHipContext * ctx, const SchroHipPackPlane * planes, int nplanes,
    int src_bpp);



/* ---- VC-2 low-delay transform data -----------------------------
 * Replaces schro_decoder_decode_lowdelay_transform_data (schrolowdelay.c:746-762):
 * the slices of a picture -- fixed-size and independent, offsets by the accumulator of
 * :607-631 -- are unpacked (interleaved exp-Golomb, schrounpack.c:211-241) and
 * dequantised (schro_dequantise, schroutils.c:180-189) one thread per slice straight
 * into the interleaved coefficient frame (sub-bands by schro_subband_get_frame_data,
 * schroparams.c:319-352; slice rectangles by schro_frame_data_get_codeblock,
 * schroframe.c:1865-1884), then the three LL bands are DC-predicted
 * (schro_decoder_subband_dc_predict (_s32), schrodecoder.c:3219-3277).  The host hands
 * over the compressed slice bytes (picture->lowdelay_buffer) instead of 2 or 4 bytes
 * per coefficient.
 *
 * Which arithmetic a picture gets follows the reference's own choice (:746-762):
 * bytes_per_sample 4 -> the s32 decoder; 2 with the chroma LL band divisible by the
 * slice counts -> the "fast" decoder, whose dequantisation is 16-bit throughout
 * (orc_dequantise_var_s16_ip, factor and offset truncated to int16_t, :478-479) and
 * whose slice_y_length field is sized from the short slice (:577); 2 otherwise -> the
 * "slow" decoder (int arithmetic, field sized per slice).  Where the reference is
 * undefined (a base index above 59 indexes past the fast decoder's tables; a corrupt
 * slice_y_length at the end of the buffer reads out of bounds) this library clamps the
 * quantiser index as the slow decoder does and reads guard bits. */
#define SCHRO_HIP_LIMIT_SUBBANDS 19     /* schrolimits.h */
typedef struct {
  int transform_depth;
  int iwt_luma_width, iwt_luma_height;
  int iwt_chroma_width, iwt_chroma_height;
  int n_horiz_slices, n_vert_slices;
  int slice_bytes_num, slice_bytes_denom;
  int quant_matrix[SCHRO_HIP_LIMIT_SUBBANDS];
} SchroHipLowDelayParams;

typedef struct {
  const uint8_t *slices;        /* device: the picture's slices back to back */
  size_t slices_bytes;          /* < 2^28 */
  void *comp[3];                /* device: Y, U, V coefficient planes (transform_frame) */
  int stride[3];                /* bytes */
} SchroHipLowDelayPicture;

/* the reference decoder a picture takes, from params and sample size (schrolowdelay.c:746-762) */
#define SCHRO_HIP_LOWDELAY_FAST16 0
#define SCHRO_HIP_LOWDELAY_SLOW16 1
#define SCHRO_HIP_LOWDELAY_S32 2
int schro_hip_lowdelay_arith (const SchroHipLowDelayParams * params, int bytes_per_sample);

/* all pictures share `params`; bytes_per_sample 2 (s16) or 4 (s32) */
int schro_hip_lowdelay_batch (SchroHipContext * ctx,
    const SchroHipLowDelayPicture * pictures, int npictures,
    const SchroHipLowDelayParams * params, int bytes_per_sample);

/* the DC prediction alone, in place on an LL band (also the intra DC prediction of the
 * core syntax, schrodecoder.c:3630-3636) */
typedef struct {
  void *data;
  int stride;
  int width, height;
} SchroHipDcPlane;
int schro_hip_dc_predict_batch (SchroHipContext * ctx,
    const SchroHipDcPlane * planes, int nplanes, int bytes_per_sample);

/* schro_frame_shift_right (schroframe.c:1265-1293) on planes, in place:
 * x = (x + ((1 << shift) >> 1)) >> shift.  The decoder applies it to an intra picture's frame
 * when the stream's bit depth exceeds the output picture's (schrodecoder.c:2013-2019). */
int schro_hip_shift_right_batch (SchroHipContext * ctx, const SchroHipDcPlane * planes, int nplanes,
    int bytes_per_sample, int shift);


/* Half-pel upsampling of one u8 component; replaces
 * schro_upsampled_frame_upsample (schroframe.c:2000-2030) /
 * schro_upsampled_gpuframe_upsample.  The device holds the reference's four planes of an
 * upsampled component -- half-pel sample (X, Y), 0 <= X < 2 * width, 0 <= Y < 2 * height, lives in
 *   plane (X & 1) + 2 * (Y & 1)   (0 integer pel, 1 h-half, 2 v-half, 3 hv-half)
 * at column X >> 1, row Y >> 1 -- in ONE buffer, tiled for the OBMC gather (r03 layout):
 *
 *   - a plane row is cut into 32-byte chunks that advance by 16 columns, so every column is
 *     stored twice and any run of up to 17 samples (one row of a block up to 16 wide, plus the
 *     X + 1 tap) starts in some chunk and ends in the same one: a lane fetches it with one
 *     byte-aligned load and needs no alignment or even / odd split instructions;
 *   - one 128-byte cache line = the same chunk of 4 consecutive rows of one plane; the four
 *     planes' lines of a (band of 4 rows, chunk) are adjacent, so the other taps of a quarter- or
 *     eighth-pel position are +-128 / +-256 bytes away and a tap the position does not use is
 *     never fetched;
 *   - 32 replicated columns in front of column 0 and behind column width - 1 (get_block clamps a
 *     block's origin to 32 pixels outside the picture, schromotion8.c:329-330), with the
 *     reference's own apron sources (schroframe.c:2012-2029: planes 0 and 1 repeat plane 0's edge
 *     sample, planes 2 and 3 plane 2's) = the half-pel column clamped to [0, 2 * width - 2]; rows
 *     have no aprons, the kernels clamp the half-pel row to [0, 2 * height - 2].
 *
 *   offset (X, Y) = (y >> 2) * stride + (xp >> 4) * 512 + plane * 128 + (y & 3) * 32 + (xp & 15)
 *   with y = Y >> 1, xp = (X >> 1) + 32; the second home of the column is 16 bytes into the chunk
 *   before: offset - 512 + 16.  stride = bytes per band of 4 rows = 512 * ((width + 79) / 16 + 1);
 *   the buffer holds ceil (height / 4) bands and must be 128-byte aligned.
 * schro_hip_upsampled_bytes () gives stride and size, schro_hip_upsampled_download () copies the
 * half-pel image to the host in linear order.  Plain (not upsampled) frames are linear.
 *
 * r04 -- PAIR images.  The U and V components of a 4:2:0 / 4:2:2 picture have the same blocks, motion
 * vectors and sample windows; a pair image holds both in the layout above over samples of TWO bytes
 * (U, V): byte column 2 * xp + c of a plane row (c = 0 U, 1 V), i.e. a chunk is 16 samples wide and
 * advances by 8, the aprons are 64 bytes:
 *   offset (X, Y, c) = (y >> 2) * stride + (xb >> 4) * 512 + plane * 128 + (y & 3) * 32 + (xb & 15),
 *   xb = 2 * ((X >> 1) + 32) + c; stride = 512 * ((2 * width + 143) / 16 + 1).
 * One load per tap then brings a block row of both components (schro_hip_obmc_batch: ref_pair).
 * schro_hip_upsample_batch writes a pair image when src_v is set (src = the U plane);
 * schro_hip_upsampled_pair_bytes / _pair_download are the helpers. */
typedef struct {
  const uint8_t *src;
  int src_stride;
  uint8_t *dst;                 /* the four tiled planes, see above */
  int dst_stride;               /* bytes per band of 4 rows, from schro_hip_upsampled_bytes / _pair_bytes */
  int width;
  int height;
  const uint8_t *src_v;         /* NULL: one component.  Else: src / src_v are the U / V planes (same size), dst a pair image */
  int src_v_stride;
} SchroHipUpsamplePlane;

/* bytes to allocate for the half-pel planes of a width x height component; *stride
 * receives the band pitch */
size_t schro_hip_upsampled_bytes (int width, int height, int *stride);
/* ... for the pair image of two width x height components */
size_t schro_hip_upsampled_pair_bytes (int width, int height, int *stride);
/* half-pel planes (on the device) -> linear host rows of 2 * width samples, 2 * height of them */
int schro_hip_upsampled_download (SchroHipContext * ctx, void *host, int host_stride,
    const void *dev, int dev_stride, int width, int height);
/* ... of a pair image: both components, each as above */
int schro_hip_upsampled_pair_download (SchroHipContext * ctx, void *host_u, void *host_v, int host_stride,
    const void *dev, int dev_stride, int width, int height);

int schro_hip_upsample_batch (SchroHipContext * ctx,
    const SchroHipUpsamplePlane * planes, int nplanes);

/* OBMC prediction + residual add + clamp for one component of one picture;
 * replaces schro_motion_render (..., add=TRUE, output_frame)
 * (schromotion.c:95 -> schro_motion_render_u8 schromotion8.c:700-929).
 *
 * mvs: DEVICE copy of the picture's SchroMotionVector array
 *      (schromotion.h:20-37, 20-byte records, x_num_blocks*y_num_blocks).
 * Block geometry is the LUMA one; the kernel derives chroma geometry from
 * chroma_h_shift/chroma_v_shift when component > 0 exactly as
 * schromotion8.c:730-758 does.
 * ref1/ref2: mv_precision == 0 -> plain u8 planes (width x height);
 *            mv_precision >= 1 -> tiled half-pel planes (see above), strides = their band pitch.
 *            ref2 may be NULL when no block uses it.
 * ref_pair:  (mv_precision >= 1, component 1 or 2) ref1 / ref2 are PAIR images of the picture's U and V
 *            components.  The U and the V plane of a picture given next to each other (U first) with the
 *            same pair images are predicted together -- one fetch per tap for both; any other use
 *            reads the component's bytes out of the pair image (correct, slower). */
typedef struct {
  const void *mvs;
  int x_num_blocks, y_num_blocks;
  int xblen_luma, yblen_luma, xbsep_luma, ybsep_luma;
  int mv_precision;
  int picture_weight_bits, picture_weight_1, picture_weight_2;
  int chroma_h_shift, chroma_v_shift;
  int component;                /* 0,1,2: selects dc[k] and chroma scaling */
  const uint8_t *ref1;
  int ref1_stride;
  const uint8_t *ref2;
  int ref2_stride;
  const void *residual;         /* s16 or s32 plane at picture coordinates */
  int residual_stride;
  int residual_bpp;             /* 2 or 4 */
  uint8_t *out;
  int out_stride;
  int width;                    /* component picture size */
  int height;
  int ref_pair;                 /* 0: one component per reference image; 1: pair images */
  int prediction_only;          /* r04: residual must be NULL; `out` receives the PREDICTION (acc + 32) >> 6 for the
                                 * combine form of schro_hip_iiwt_batch.  The prediction must fit 8 bits -- it does for
                                 * every legal stream: weights with picture_weight_1, _2 >= 0 and a sum <= 1 << bits are
                                 * required (else an error at the call), and a DC value outside [-128, 127] makes the
                                 * launch raise a flag: the next synchronising call of the context answers
                                 * SCHRO_HIP_ENEEDS_RESIDUAL and names the batch (such pictures need the
                                 * residual form, which follows the reference's 16-bit wrap-around).
                                 * r06, 2: residual NULL, `out` is an int16 plane that receives (acc - 8160) >> 6 =
                                 * the prediction - 128 in the reference's 16-bit arithmetic (orc_rrshift6_s16_ip_2d,
                                 * schroorc.orc:676-682: what schro_motion_render (add = FALSE) leaves in its dest and
                                 * schro_motion_render_cuda's contract): any weights, any DC values. */
} SchroHipObmcPlane;

int schro_hip_obmc_batch (SchroHipContext * ctx,
    const SchroHipObmcPlane * planes, int nplanes);
/* r05: prediction_only calls of schro_hip_obmc_batch are numbered per context (1, 2, ...); this is the number of
 * the latest one (0: none yet) -- what a later SCHRO_HIP_ENEEDS_RESIDUAL names. */
unsigned int schro_hip_obmc_prediction_epoch (SchroHipContext * ctx);
/* r06: the numbers of the FINISHED prediction_only batches whose predictions did not fit 8 bits and that this call has
 * not returned before, oldest first: up to `max` of them into epochs[], the count as the result (0: none; < 0: an
 * error).  Nothing is lost between calls: a batch's flag is read once its launches have completed and kept until it
 * has been returned here; a synchronising call names the same batches once in its SCHRO_HIP_ENEEDS_RESIDUAL status
 * (batches already returned here are not named again).  A host may run any number of prediction_only batches ahead:
 * the thirteenth unfinished one waits for the first. */
int schro_hip_obmc_overflowed (SchroHipContext * ctx, unsigned int *epochs, int max);

/* ---- core-syntax coefficients: dequantisation on the device ----------------
 *
 * schro_decoder_decode_subband (schrodecoder.c:3525-3640) does two things per codeblock: the
 * serial entropy decode (binary arithmetic coder or VLC), and the data-parallel rest -- zero
 * fill of zero codeblocks (:3311-3322) and dequantisation (:3072-3079, :3400-3451,
 * orc_dequantise_s16_* / _s32_ip_2d schroorc.orc:1098-1219).  The arithmetic decoder's
 * contexts look only at whether neighbouring / parent coefficients are zero and at a
 * neighbour's sign, which quantised values answer as well as dequantised ones (a non-zero value
 * stays non-zero: quant_factor >= 4), so a host decoder can keep the QUANTISED values, hand
 * them over for the non-zero codeblocks only -- 1, 2 or 4 bytes each -- and leave the dense
 * coefficient frame to the device: schro_hip_dequant_batch, then schro_hip_dc_predict_batch on
 * the LL bands of intra pictures (schrodecoder.c:3629-3636), then schro_hip_iiwt_batch. */
typedef struct {
  int dst_offset;               /* bytes from the plane's base to the codeblock's first sample */
  int dst_stride;               /* bytes between its rows: the frame stride << the sub-band's level shift
                                 * (schro_subband_get_frame_data, schroparams.c:319-368) */
  int width, height;            /* samples: xmax - xmin, ymax - ymin */
  int src_offset;               /* bytes from `values` to its quantised values, row-major and
                                 * tight; < 0: zero codeblock (nothing stored) */
  unsigned char src_bytes;      /* 1, 2 or 4 bytes per stored value (signed) */
  unsigned char quant_index;    /* ctx->quant_index of the codeblock, 0 .. 60 */
  unsigned char pad[2];
} SchroHipCodeblock;

typedef struct {
  void *dst;                    /* the component's coefficient plane (device, s16 or s32) */
  const void *values;           /* packed quantised values of its non-zero codeblocks (device) */
  const SchroHipCodeblock *codeblocks;  /* HOST array: every codeblock of every sub-band */
  int ncodeblocks;
  int is_intra;                 /* params->num_refs == 0: schro_table_offset_1_2, else _3_8 */
} SchroHipDequantPlane;

/* arith 0: C int arithmetic (arithmetic-coded codeblocks; s32 frames); 1: the 16-bit Orc
 * arithmetic of the VLC (is_noarith) path on s16 frames.  The two differ only where 16-bit
 * products wrap, which no legal stream reaches. */
int schro_hip_dequant_batch (SchroHipContext * ctx, const SchroHipDequantPlane * planes, int nplanes,
    int bytes_per_sample, int arith);

/* r04 -- plans: the host cost of a repeated picture geometry is O (planes), not O (codeblocks).  A plan holds
 * what is fixed per picture GEOMETRY -- every record's dst_offset / dst_stride / width / height and with them the
 * launch's tiles -- on the device; a run uploads the records as they are (the device reads src_offset, src_bytes
 * and quant_index itself) and one line per plane (dst, values, is_intra).  `planes` of a run: the same planes in
 * the same order, with this batch's pointers and records; the records' geometry must be the plan's.  Results
 * are schro_hip_dequant_batch's, bit for bit.  (r03: the batch call rebuilt 15 k job records per 8 x 2160p, 1.3 of
 * the 2.3 ms of a PCIe-inclusive step with the quantised hand-over.) */
typedef struct SchroHipDequantPlan SchroHipDequantPlan;
SchroHipDequantPlan *schro_hip_dequant_plan_new (SchroHipContext * ctx, const SchroHipDequantPlane * planes, int nplanes,
    int bytes_per_sample, int arith);
int schro_hip_dequant_plan_run (SchroHipDequantPlan * plan, const SchroHipDequantPlane * planes, int nplanes);
void schro_hip_dequant_plan_free (SchroHipDequantPlan * plan);

/* The geometry of every codeblock record of one component, in the decoder's order -- sub-band index
 * 0 .. 3 * depth, in each its rows of codeblocks (schro_decoder_decode_subband, schrodecoder.c:3558-3577;
 * counts per sub-band from params->horiz_codeblocks / vert_codeblocks [0 .. depth] as
 * schro_decoder_setup_codeblocks picks them, :3280-3293; rectangles by schro_subband_get_frame_data,
 * schroparams.c:319-352): dst_offset, dst_stride, width, height filled, src_offset -1 (zero codeblock).
 * A host decoder builds the table once per picture geometry and fills src_offset / src_bytes /
 * quant_index as it entropy-decodes.  Returns the number of records (write stops at `max`). */
int schro_hip_codeblock_layout (int iwt_width, int iwt_height, int transform_depth, const int *horiz_codeblocks,
    const int *vert_codeblocks, int stride, int bytes_per_sample, SchroHipCodeblock * out, int max);

/* ---- frame layer: the reference's stage boundary ------------------------- */

/* The structs of this layer are LAYOUT-IDENTICAL to the reference's (same members, same
 * order, same offsets on LP64): a SchroFrame * / SchroParams * / SchroMotion * /
 * SchroMemoryDomain * of schroedinger 1.0.11 can be passed where this header says
 * SchroHipFrame * / SchroHipParams * / SchroHipMotion * / SchroHipMemoryDomain *.  The offsets
 * below were recorded from the reference's own headers by scripts/ref_layout.py
 * (tests/golden/ref_layout.json) and are pinned here by _Static_assert; where the
 * reference's headers are present, tests/c/layout_check.c compares member by member. */

#define SCHRO_HIP_FRAME_CACHE_SIZE 32   /* SCHRO_FRAME_CACHE_SIZE, schroframe.h:56 */
#define SCHRO_HIP_LIMIT_TRANSFORM_DEPTH 6       /* schrolimits.h:50 */
#define SCHRO_HIP_LIMIT_BLOCK_SIZE 64   /* schrolimits.h:67 */
#define SCHRO_HIP_MEMORY_DOMAIN_SLOTS 1000      /* schrodomain.h:11 */
/* schrodomain.h:34-36 defines CPU 0x1, CUDA 0x2, OPENGL 0x4; the next free bit */
#define SCHRO_MEMORY_DOMAIN_HIP 0x0008

/* SchroMemoryDomain (schrodomain.h:13-28): the reference's slot cache
 * (schro_memory_domain_alloc, schrodomain.c:58-137) calls alloc / free of this table, so
 * schro_frame_new_and_alloc (domain, ...) (schroframe.c:172-188) hands out device frames
 * when `domain` came from schro_memory_domain_new_hip.  alloc / free take no domain argument
 * (the reference's signature): they use the HIP domain of the calling thread's current
 * device.  The members after `slots` are private to this library. */
typedef struct _SchroHipMemoryDomain {
  void *mutex;                  /* SchroMutex *: the reference host creates and owns it */
  unsigned int flags;           /* SCHRO_MEMORY_DOMAIN_HIP */
  void *(*alloc) (int size);
  void *(*alloc_2d) (int depth, int width, int height);
  void (*free) (void *ptr, int size);
  struct {
    unsigned int flags;
    void *ptr;
    int size;
    void *priv;
  } slots[SCHRO_HIP_MEMORY_DOMAIN_SLOTS];
  /* -- end of the reference's struct -- */
  SchroHipContext *ctx;
} SchroHipMemoryDomain;

/* schro_memory_domain_new_cuda (schrocuda.h:9) replacement: a context on `device` (its queues,
 * caches) behind a SchroMemoryDomain-shaped handle.  schro_hip_domain_context gives the context
 * for the plane-layer calls; schro_memory_domain_free_hip releases both. */
SchroHipMemoryDomain *schro_memory_domain_new_hip (int device);
void schro_memory_domain_free_hip (SchroHipMemoryDomain * domain);
/* a domain of PINNED HOST memory (flags SCHRO_MEMORY_DOMAIN_CPU): see schro_hip_host_alloc; free it with free () */
SchroHipMemoryDomain *schro_memory_domain_new_hip_host (void);
SchroHipContext *schro_hip_domain_context (SchroHipMemoryDomain * domain);
/* the domain of a context made with schro_hip_context_new */
SchroHipMemoryDomain *schro_hip_context_domain (SchroHipContext * ctx);

/* SchroFrameData (schroframe.h:58-67) */
typedef struct {
  int format;                   /* SchroFrameFormat */
  void *data;
  int stride;
  int width;
  int height;
  int length;
  int h_shift;
  int v_shift;
} SchroHipFrameData;

/* SchroFrame (schroframe.h:69-94).  domain == NULL: host memory; a HIP domain: `data` of
 * every component is a device pointer.  An upsampled device frame (is_upsampled) holds the
 * tiled half-pel images in its components and keeps its integer-pel source frame in
 * virt_frame1. */
typedef struct _SchroHipFrame SchroHipFrame;
struct _SchroHipFrame {
  int refcount;
  void (*free) (SchroHipFrame * frame, void *priv);
  SchroHipMemoryDomain *domain;
  void *regions[3];
  void *priv;

  int format;
  int width;
  int height;

  SchroHipFrameData components[3];

  int is_virtual;
  int cached_lines[3][SCHRO_HIP_FRAME_CACHE_SIZE];
  SchroHipFrame *virt_frame1;
  SchroHipFrame *virt_frame2;
  void (*render_line) (SchroHipFrame * frame, void *dest, int component, int i);
  void *virt_priv;
  void *virt_priv2;

  int extension;
  int cache_offset[3];
  int is_upsampled;
  int upsample_done;            /* schro_bool */
};

/* SchroGlobalMotion, SchroParams (schroparams.h:18-74) */
typedef struct {
  int b0, b1, a_exp, a00, a01, a10, a11, c_exp, c0, c1;
} SchroHipGlobalMotion;

typedef struct {
  void *video_format;           /* SchroVideoFormat * */
  int is_noarith;
  int wavelet_filter_index;
  int transform_depth;
  int horiz_codeblocks[SCHRO_HIP_LIMIT_TRANSFORM_DEPTH + 1];
  int vert_codeblocks[SCHRO_HIP_LIMIT_TRANSFORM_DEPTH + 1];
  int codeblock_mode_index;
  int num_refs;
  int have_global_motion;
  int xblen_luma;
  int yblen_luma;
  int xbsep_luma;
  int ybsep_luma;
  int mv_precision;
  SchroHipGlobalMotion global_motion[2];
  int picture_pred_mode;
  int picture_weight_bits;
  int picture_weight_1;
  int picture_weight_2;
  int is_lowdelay;
  int n_horiz_slices;
  int n_vert_slices;
  int slice_bytes_num;
  int slice_bytes_denom;
  int quant_matrix[3 * SCHRO_HIP_LIMIT_TRANSFORM_DEPTH + 1];
  int iwt_chroma_width;
  int iwt_chroma_height;
  int iwt_luma_width;
  int iwt_luma_height;
  int x_num_blocks;
  int y_num_blocks;
  int x_offset;
  int y_offset;
} SchroHipParams;

/* SchroMotion (schromotion.h:53-86); schro_motion_render_hip reads src1, src2,
 * motion_vectors and params, as schro_motion_render_u8 does (schromotion8.c:700-730) */
typedef struct {
  SchroHipFrame *src1;          /* device; plain u8 if mv_precision == 0 else upsampled */
  SchroHipFrame *src2;          /* may be NULL */
  void *motion_vectors;         /* HOST SchroMotionVector array (schromotion.h:20-37) */
  SchroHipParams *params;
  int ref_weight_precision;
  int ref1_weight;
  int ref2_weight;
  int mv_precision;
  int xoffset;
  int yoffset;
  int xbsep;
  int ybsep;
  int xblen;
  int yblen;
  SchroHipFrameData block;
  SchroHipFrameData alloc_block;
  SchroHipFrameData obmc_weight;
  SchroHipFrameData alloc_block_ref[2];
  SchroHipFrameData block_ref[2];
  int weight_x[SCHRO_HIP_LIMIT_BLOCK_SIZE];
  int weight_y[SCHRO_HIP_LIMIT_BLOCK_SIZE];
  int width;
  int height;
  int max_fast_x;
  int max_fast_y;
  int simple_weight;            /* schro_bool */
  int oneref_noscale;
} SchroHipMotion;

#if defined(__LP64__) || defined(_LP64)
#include <stddef.h>
#ifdef __cplusplus
#define SCHRO_HIP_LAYOUT(t, m, off) static_assert (offsetof (t, m) == (off), #t "." #m)
#define SCHRO_HIP_SIZE(t, n) static_assert (sizeof (t) == (n), #t)
#else
#define SCHRO_HIP_LAYOUT(t, m, off) _Static_assert (offsetof (t, m) == (off), #t "." #m)
#define SCHRO_HIP_SIZE(t, n) _Static_assert (sizeof (t) == (n), #t)
#endif
/* numbers: tests/golden/ref_layout.json (scripts/ref_layout.py, from the reference's headers) */
SCHRO_HIP_SIZE (SchroHipFrameData, 40);
SCHRO_HIP_LAYOUT (SchroHipFrameData, data, 8);
SCHRO_HIP_LAYOUT (SchroHipFrameData, stride, 16);
SCHRO_HIP_LAYOUT (SchroHipFrameData, v_shift, 36);
SCHRO_HIP_SIZE (SchroHipFrame, 648);
SCHRO_HIP_LAYOUT (SchroHipFrame, free, 8);
SCHRO_HIP_LAYOUT (SchroHipFrame, domain, 16);
SCHRO_HIP_LAYOUT (SchroHipFrame, regions, 24);
SCHRO_HIP_LAYOUT (SchroHipFrame, priv, 48);
SCHRO_HIP_LAYOUT (SchroHipFrame, format, 56);
SCHRO_HIP_LAYOUT (SchroHipFrame, width, 60);
SCHRO_HIP_LAYOUT (SchroHipFrame, height, 64);
SCHRO_HIP_LAYOUT (SchroHipFrame, components, 72);
SCHRO_HIP_LAYOUT (SchroHipFrame, is_virtual, 192);
SCHRO_HIP_LAYOUT (SchroHipFrame, cached_lines, 196);
SCHRO_HIP_LAYOUT (SchroHipFrame, virt_frame1, 584);
SCHRO_HIP_LAYOUT (SchroHipFrame, render_line, 600);
SCHRO_HIP_LAYOUT (SchroHipFrame, extension, 624);
SCHRO_HIP_LAYOUT (SchroHipFrame, is_upsampled, 640);
SCHRO_HIP_LAYOUT (SchroHipFrame, upsample_done, 644);
SCHRO_HIP_SIZE (SchroHipParams, 336);
SCHRO_HIP_LAYOUT (SchroHipParams, wavelet_filter_index, 12);
SCHRO_HIP_LAYOUT (SchroHipParams, transform_depth, 16);
SCHRO_HIP_LAYOUT (SchroHipParams, num_refs, 80);
SCHRO_HIP_LAYOUT (SchroHipParams, have_global_motion, 84);
SCHRO_HIP_LAYOUT (SchroHipParams, xblen_luma, 88);
SCHRO_HIP_LAYOUT (SchroHipParams, mv_precision, 104);
SCHRO_HIP_LAYOUT (SchroHipParams, picture_weight_bits, 192);
SCHRO_HIP_LAYOUT (SchroHipParams, is_lowdelay, 204);
SCHRO_HIP_LAYOUT (SchroHipParams, slice_bytes_num, 216);
SCHRO_HIP_LAYOUT (SchroHipParams, quant_matrix, 224);
SCHRO_HIP_LAYOUT (SchroHipParams, iwt_chroma_width, 300);
SCHRO_HIP_LAYOUT (SchroHipParams, iwt_luma_width, 308);
SCHRO_HIP_LAYOUT (SchroHipParams, x_num_blocks, 316);
SCHRO_HIP_LAYOUT (SchroHipParams, y_offset, 328);
SCHRO_HIP_LAYOUT (SchroHipMemoryDomain, flags, 8);
SCHRO_HIP_LAYOUT (SchroHipMemoryDomain, alloc, 16);
SCHRO_HIP_LAYOUT (SchroHipMemoryDomain, alloc_2d, 24);
SCHRO_HIP_LAYOUT (SchroHipMemoryDomain, free, 32);
SCHRO_HIP_LAYOUT (SchroHipMemoryDomain, slots, 40);
SCHRO_HIP_LAYOUT (SchroHipMemoryDomain, ctx, 32040);
SCHRO_HIP_LAYOUT (SchroHipMotion, motion_vectors, 16);
SCHRO_HIP_LAYOUT (SchroHipMotion, params, 24);
SCHRO_HIP_LAYOUT (SchroHipMotion, ref_weight_precision, 32);
#endif

/* schro_frame_new_and_alloc (schroframe.c:60-191) on the device domain:
 * planar Y,U,V, stride = round-up-64 (width * bytes); upsampled == 1
 * allocates 2w x 2h half-pel images per component. */
SchroHipFrame *schro_hip_frame_new_and_alloc (SchroHipContext * ctx,
    int format, int width, int height, int upsampled);
SchroHipFrame *schro_hip_frame_ref (SchroHipFrame * frame);
void schro_hip_frame_unref (SchroHipFrame * frame);

/* schro_frame_to_gpu / schro_gpuframe_to_cpu (schrogpuframe.h:17-18):
 * whole-frame copies, all three components, synchronous on return. */
int schro_frame_to_hip (SchroHipFrame * dest, SchroHipFrame * src);
int schro_hipframe_to_cpu (SchroHipFrame * dest, SchroHipFrame * src);
/* r03: the same, enqueued on the context's selected queue and NOT waited for (see the asynchronous
 * transfers above: host frames in pinned memory, queues 2 / 3, marks, schro_hip_queue_synchronize) */
int schro_frame_to_hip_async (SchroHipFrame * dest, SchroHipFrame * src);
int schro_hipframe_to_cpu_async (SchroHipFrame * dest, SchroHipFrame * src);
/* a copy of a device frame (plain or upsampled) on another context's device: one hipMemcpyPeerAsync per
 * component on dst_ctx's selected queue, complete on return (the scheduler's reference migration) */
SchroHipFrame *schro_hip_frame_copy_to (SchroHipContext * dst_ctx, SchroHipFrame * src);

/* How the stage calls below end.  complete_on_return != 0 (the default): the reference's contract -- a stage
 * is complete when its function returns (schroasync-pthread.c:320-328); the call waits for the selected
 * queue.  0: the calls only ENQUEUE on the selected queue (frames must be on the device already; the
 * motion vectors go through pinned staging buffers), for a host that keeps several pictures in flight and
 * orders them with marks (0.58 ms per 2160p picture, host hand-over included, against 0.98 ms
 * under the contract). */
int schro_hip_context_set_stage_completion (SchroHipContext * ctx, int complete_on_return);

/* schro_frame_inverse_iwt_transform_cuda (schrocuda.h:13-14) replacement, same arguments:
 * upload transform_frame (host) or use it where it is (device), run the multi-level inverse
 * transform into `frame` (device, iwt-padded size). */
int schro_frame_inverse_iwt_transform_hip (SchroHipFrame * frame,
    SchroHipFrame * transform_frame, SchroHipParams * params);
/* r04 -- the inverse transform and x_combine's add in one call: output_frame (device u8) = sat_u8 (inverse
 * transform of transform_frame + prediction), prediction = the frame schro_motion_render_hip (motion, dest, NULL,
 * FALSE, NULL) rendered (the mc_tmp_frame of the reference's GPU paths, schrodecoder.c:1742-1760, :1908-1921), or
 * NULL for a picture without references (+ 128, :1788-1790).  The residual picture never exists in memory. */
int schro_frame_inverse_iwt_transform_combine_hip (SchroHipFrame * output_frame, SchroHipFrame * transform_frame,
    SchroHipParams * params, SchroHipFrame * prediction);
/* r05: the same for a picture WITHOUT references whose output picture is v210 (BASELINE config 5): x_wavelet_transform and
 * the schro_frame_convert of x_combine (schrodecoder.c:2011-2052) in one call -- `packed` a device frame of format v210,
 * transform_frame a device s16 / s32 4:2:2 frame; schro_hipframe_to_cpu (output_picture, packed) follows.  Same bytes as
 * schro_frame_inverse_iwt_transform_hip + schro_hipframe_convert; the pixel frame is not written where
 * schro_hip_iiwt_pack_v210_batch's fused kernel applies. */
int schro_frame_inverse_iwt_transform_convert_hip (SchroHipFrame * packed, SchroHipFrame * transform_frame, SchroHipParams * params);

/* schro_decoder_decode_lowdelay_transform_data (picture), schrolowdelay.c:746-762, with
 * picture->transform_frame on the device: `slices` is picture->lowdelay_buffer->data (host),
 * copied to the device compressed; the frame's format picks s16 / s32.  Synchronous. */
int schro_hip_decode_lowdelay_transform_data (SchroHipFrame * transform_frame,
    const void *slices, size_t slices_bytes, const SchroHipLowDelayParams * params);

/* r05 -- the quantised hand-over behind the frame layer (VERDICT r04 "missing" 1): what a patched
 * schro_decoder_decode_subband (schrodecoder.c:3525-3640) leaves behind for one picture instead of a dense
 * coefficient frame.  Per component: the codeblock records in the decoder's order (their geometry from
 * schro_hip_codeblock_layout with the DEVICE transform frame's stride; src_offset / src_bytes / quant_index filled
 * in as schro_decoder_decode_codeblock goes, src_offset -1 for a zero codeblock, :3311-3322) and the quantised
 * values of its non-zero codeblocks, row-major and tight, 1 / 2 / 4 bytes each (src_offset counts from the
 * component's `values`).  INTEGRATION 3 shows the #ifdef HAVE_HIP branch. */
typedef struct {
  const SchroHipCodeblock *codeblocks[3];       /* HOST arrays (read during the call) */
  int ncodeblocks[3];
  const void *values[3];        /* values_on_device: device pointers (the host uploaded the picture's blob on the copy queue,
                                 * INTEGRATION 3a); else host pointers, copied by the call (pinned memory if the copy is to be asynchronous) */
  size_t values_bytes[3];
  int values_on_device;
} SchroHipQuantisedPicture;

/* transform_frame: the picture's DEVICE transform frame (s16 or s32), filled completely: zero codeblocks are zero-filled,
 * the others dequantised with the arithmetic of the decoder the picture takes (params->is_noarith on an s16 frame: the
 * 16-bit Orc program of the VLC path, schroorc.orc:1098-1219; else C int, schrodecoder.c:3072-3079), and for a picture
 * without references (params->num_refs == 0) the LL band of every component is DC-predicted (:3629-3636, :3219-3277).
 * What follows is schro_frame_inverse_iwt_transform_hip (frame, transform_frame, params) as before.  The launch
 * geometry is kept in a plan of the context's (schro_hip_dequant_plan_*), rebuilt when the picture geometry changes.
 * Enqueued on the selected queue; complete on return unless stage completion is off. */
int schro_hipframe_dequantise (SchroHipFrame * transform_frame, const SchroHipQuantisedPicture * quantised,
    SchroHipParams * params);

/* schro_upsampled_gpuframe_upsample (schrogpuframe.h:27) replacement:
 * dest (device, is_upsampled) <- half-pel images of src (device u8). */
int schro_upsampled_hipframe_upsample (SchroHipFrame * dest, SchroHipFrame * src);
/* the reference's one-argument form (schrogpuframe.h:29): `frame` is the upsampled device frame, its
 * integer-pel source the frame it keeps in virt_frame1 */
int schro_upsampled_hipframe_upsample_inplace (SchroHipFrame * frame);

/* schro_motion_render (motion, dest, addframe, add, output_frame) (schromotion.h:100, as
 * x_render_motion calls it, schrodecoder.c:1905-1935) replacement, same arguments.  add TRUE: the CPU path's
 * fused form -- `dest` (its s16 scratch frame) is not used and may be NULL.  add FALSE (r04): the prediction alone
 * into `dest`, a u8 device frame (schro_motion_render_cuda (motion, mc_tmp_frame), :1759), addframe and output_frame
 * NULL -- for schro_frame_inverse_iwt_transform_combine_hip; DC values outside [-128, 127] are an error there.  addframe:
 * device s16/s32 residual (picture->frame), or NULL for a zero_residual picture (nothing is added,
 * nothing is read: schrodecoder.c:1904-1906); output_frame: device u8.  motion->motion_vectors: the
 * host array, or a device copy of it.  Global motion is not
 * supported (the reference routes it to a different renderer, schromotion.c:113-118)
 * -> SCHRO_HIP_EUNSUPPORTED.
 * r06 -- add FALSE with an S16 device frame as `dest`: the literal contract of schro_motion_render_cuda (motion,
 * mc_tmp_frame) (schrocuda.h:13; schrodecoder.c:1742-1760, where mc_tmp_frame is an S16 frame of the transform's padded
 * size): dest receives the prediction - 128 in the reference's 16-bit arithmetic ((acc - 8160) >> 6,
 * orc_rrshift6_s16_ip_2d / _sub_s16_2d's d2) over the area the references cover; any weights, any DC values, never
 * SCHRO_HIP_ENEEDS_RESIDUAL.  schro_hipframe_add (picture->frame, mc_tmp_frame) and schro_hipframe_convert (output,
 * picture->frame) then finish the picture exactly as the HAVE_CUDA branch does (:1908-1910, :2011). */
int schro_motion_render_hip (SchroHipMotion * motion, SchroHipFrame * dest,
    SchroHipFrame * addframe, int add, SchroHipFrame * output_frame);

/* schro_gpuframe_add (dest, src) (schrogpuframe.h:18; call site schrodecoder.c:1908-1910) = schro_frame_add
 * (schroframe.c:1000-1029) on device frames: dest (S16) += src (S16, or U8 zero-extended) of the same chroma format,
 * 16-bit wrapping add over the components' common size.  Other depth pairs: SCHRO_HIP_EINVAL (the reference asserts). */
int schro_hipframe_add (SchroHipFrame * dest, SchroHipFrame * src);

/* schro_gpuframe_convert (schrogpuframe.h:20) replacement for the conversions the decode path
 * performs: s16/s32 -> u8 (+128, clamp, crop), u8 -> u8 copy, planar -> packed (YUYV, UYVY,
 * AYUV from u8; v210, v216, ARGB, AY64 from u8 / s16 / s32). */
int schro_hipframe_convert (SchroHipFrame * dest, SchroHipFrame * src);
/* schro_frame_shift_right (schroframe.c:1265-1293) on a device s16 / s32 frame */
int schro_hipframe_shift_right (SchroHipFrame * frame, int shift);

/* diagnostics: with SCHRO_HIP_OBMC_STAMPS set, the row / staged OBMC kernels record cycle stamps
 * per phase and workgroup; this prints their medians to stderr (scratch runs only) */
void schro_hip_obmc_stamps_dump (void);

/* ---- several devices behind one decode loop -----------------------------
 *
 * One exec-domain thread per device, each with its own context -- the reference's
 * schro_async_add_exec_domain thread (schroasync-pthread.c:362-390), N times -- and the
 * device-affinity rule the reference's scheduler lacks (schro_decoder_async_schedule,
 * schrodecoder.c:1546-1682; schro_picture_new :332-400): a picture runs on the device that
 * holds its first reference, so a reference chain (closed GOP) never leaves its device; a
 * picture without references starts a chain on the least loaded device (ties: the device holding the fewest live references).  A device runs its
 * pictures in submission (= coded) order, which puts references before their dependents.
 * `func` is the picture's pixel path -- the stage calls of this header on `ctx` -- and runs on
 * that device's thread; no collective, no peer traffic unless a picture predicts across two
 * chains (then *foreign_ref names the reference that lives elsewhere, the picture waits for
 * it, and moving that frame is the caller's: one u8 picture over xGMI). */
typedef struct SchroHipScheduler SchroHipScheduler;
typedef int (*SchroHipPictureFunc) (SchroHipContext * ctx, int device_index, void *priv);

/* n_devices 0: every visible device.  _virtual: threads and affinity without devices (ctx is
 * NULL in the callbacks): the host logic under test on a machine without GPUs. */
SchroHipScheduler *schro_hip_scheduler_new (int n_devices);
SchroHipScheduler *schro_hip_scheduler_new_virtual (int n_devices);
/* an explicit device list; a device may appear more than once (two exec-domain threads and contexts
 * on one device: how a box with one GPU exercises the cross-device paths) */
SchroHipScheduler *schro_hip_scheduler_new_on (const int *devices, int n_devices);
/* waits for everything submitted, stops the threads, frees the contexts */
void schro_hip_scheduler_free (SchroHipScheduler * sched);
int schro_hip_scheduler_n_devices (SchroHipScheduler * sched);
SchroHipContext *schro_hip_scheduler_context (SchroHipScheduler * sched, int device_index);
/* returns the device index the picture was given (< 0: error).  refs: the picture numbers it
 * predicts from (0 .. 2), each submitted earlier with is_ref != 0. */
int schro_hip_scheduler_submit (SchroHipScheduler * sched, int picture_number, const int *refs, int n_refs,
    int is_ref, SchroHipPictureFunc func, void *priv, int *foreign_ref);
/* A reference picture leaves the reference queue (schro_decoder_reference_retire,
 * schrodecoder.c:1302 -- at PARSE time, possibly before pictures that predict from it, or the
 * reference itself, have run): later submits no longer find the number; pictures already submitted
 * keep what they resolved, and the reference's frames go when the last of them has finished. */
int schro_hip_scheduler_retire (SchroHipScheduler * sched, int picture_number);
/* r03 -- references move, not just wait.  The function of a reference picture publishes the device
 * frame its dependents read (the upsampled frame when mv_precision > 0, else the plain one); the
 * scheduler keeps a reference on it.  Before a picture with a reference on ANOTHER device runs, that
 * frame is copied to the picture's device (schro_hip_frame_copy_to: one hipMemcpyPeerAsync per
 * component; the owner's device work is complete by then) and cached there for later dependents;
 * schro_hip_scheduler_reference_frame, called by a picture's function, gives the frame of one of ITS
 * references on ITS device -- the published frame or the copy (NULL: nothing was published: the
 * foreign_ref of submit names what the caller has to move itself).  On virtual devices the
 * "frames" are opaque pointers that are handed through. */
int schro_hip_scheduler_publish_reference (SchroHipScheduler * sched, int device_index, void *frame);
void *schro_hip_scheduler_reference_frame (SchroHipScheduler * sched, int device_index, int picture_number);
long schro_hip_scheduler_moves (SchroHipScheduler * sched);       /* frames copied between devices so far */
/* r04: nothing on the path drains a device.  A reference picture is complete when an event stands behind
 * the work its function enqueued; pictures of the same device follow it in the in-order KERNEL queues (r05:
 * both of them wait for that event, and every picture function starts with queue 0 selected; what a function
 * puts on the COPY queues it orders itself with marks), a picture
 * on another device waits for the event on its copy queue and copies asynchronously (TODO-CUDA:5-7) -- the
 * scheduler waits for nothing itself, and the peer copy does not hold its caller either (measured: 22 us in the
 * call behind an unfired event, two contexts on one device, profiles/r05_peer_copy_hip_trace.txt; unlike copies from /
 * to pinned host memory; a copy between two real devices is unmeasured).
 * r05: a frame the scheduler lets go of (a retired reference, its copies on other devices) is released only
 * when the kernel queues of its device have passed that point -- kernels of dependents may still read it.
 * A reference whose function returned an error is marked failed: its dependents -- and theirs -- do not
 * run, they finish with SCHRO_HIP_ESKIPPED (the reference decoder skips such pictures: picture->error,
 * schrodecoder.c:1308-1311, :1399-1418); schro_hip_scheduler_wait still reports the first real error. */
long schro_hip_scheduler_skipped (SchroHipScheduler * sched);     /* pictures skipped so far */
/* most reference pictures of ONE device whose device work was still running when the next one had been enqueued */
int schro_hip_scheduler_refs_in_flight_max (SchroHipScheduler * sched);
/* waits until every submitted picture has run; returns the first non-zero result of a func */
int schro_hip_scheduler_wait (SchroHipScheduler * sched);

#ifdef __cplusplus
}
#endif
#endif
