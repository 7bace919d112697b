/* A C caller of the frame layer (include/schro_hip.h), replaying the reference decoder's
 * pixel-path stages for one inter picture in the order schrodecoder.c runs them:
 *
 *   schro_decoder_x_wavelet_transform  (schrodecoder.c:1809-1853)  -> schro_frame_inverse_iwt_transform_hip
 *   schro_decoder_x_upsample           (schrodecoder.c:1697-1727)  -> schro_upsampled_hipframe_upsample
 *   schro_decoder_x_render_motion      (schrodecoder.c:1905-1935)  -> schro_motion_render_hip
 *   schro_decoder_x_combine            (schrodecoder.c:1937-2141)  -> schro_hipframe_convert, schro_hipframe_to_cpu
 *
 * with frames that are SchroFrame-shaped structs (the header's mirror types are layout-identical
 * to the reference's: tests/test_ref_layout.py) living in a SchroMemoryDomain-shaped HIP domain.
 * Inputs come from DIR (written by tests/test_gpu_c_harness.py), the decoded picture and the
 * residual go back to DIR; the Python test compares them with the oracle.
 *
 *   gcc -Iinclude tests/c/stage_replay.c -Lschroedinger_amd -lschro_hip -o tests/c/_build/stage_replay
 *   stage_replay DIR */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "schro_hip.h"

#define CHECK(e) do { int r_ = (e); if (r_) { fprintf (stderr, "%s -> %d: %s\n", #e, r_, schro_hip_last_error ()); return 1; } } while (0)

static void *
read_file (const char *dir, const char *name, size_t bytes)
{
  char path[1024];
  snprintf (path, sizeof (path), "%s/%s", dir, name);
  FILE *f = fopen (path, "rb");
  void *p = malloc (bytes ? bytes : 1);
  if (!f || fread (p, 1, bytes, f) != bytes) {
    fprintf (stderr, "cannot read %zu bytes of %s\n", bytes, path);
    exit (2);
  }
  fclose (f);
  return p;
}

static void
write_file (const char *dir, const char *name, const void *p, size_t bytes)
{
  char path[1024];
  snprintf (path, sizeof (path), "%s/%s", dir, name);
  FILE *f = fopen (path, "wb");
  if (!f || fwrite (p, 1, bytes, f) != bytes) {
    fprintf (stderr, "cannot write %s\n", path);
    exit (2);
  }
  fclose (f);
}

/* a host SchroFrame over one contiguous buffer: planar Y, U, V, tight strides
 * (schro_frame_new_from_data_*, schroframe.c:233-330) */
static void
host_frame (SchroHipFrame * f, int format, int bpp, int w, int h, int cw, int ch, void *data)
{
  memset (f, 0, sizeof (*f));
  f->refcount = 1;
  f->format = format;
  f->width = w;
  f->height = h;
  char *p = (char *) data;
  for (int k = 0; k < 3; k++) {
    SchroHipFrameData *c = &f->components[k];
    c->format = format;
    c->width = k ? cw : w;
    c->height = k ? ch : h;
    c->stride = c->width * bpp;
    c->length = c->stride * c->height;
    c->data = p;
    c->h_shift = k ? SCHRO_HIP_FORMAT_H_SHIFT (format) : 0;
    c->v_shift = k ? SCHRO_HIP_FORMAT_V_SHIFT (format) : 0;
    p += c->length;
  }
}

/* schro_frame_new_and_alloc (domain, format, width, height) as the REFERENCE does it
 * (schro_frame_new_and_alloc_full, schroframe.c:60-191, extension 0, not upsampled): strides
 * ROUND_UP_16 (width * bytes), the three components back to back in ONE block from the domain's
 * alloc table -- the frames a patched decoder hands to the stage calls come from here, not from
 * schro_hip_frame_new_and_alloc */
static SchroHipFrame *
domain_frame (SchroHipMemoryDomain * domain, int format, int bpp, int w, int h)
{
  SchroHipFrame *f = (SchroHipFrame *) calloc (1, sizeof (SchroHipFrame));
  const int hs = SCHRO_HIP_FORMAT_H_SHIFT (format), vs = SCHRO_HIP_FORMAT_V_SHIFT (format);
  const int cw = (w + (1 << hs) - 1) >> hs, ch = (h + (1 << vs) - 1) >> vs;
  int total = 0;
  f->refcount = 1;
  f->domain = domain;
  f->format = format;
  f->width = w;
  f->height = h;
  for (int k = 0; k < 3; k++) {
    SchroHipFrameData *c = &f->components[k];
    c->format = format;
    c->width = k ? cw : w;
    c->height = k ? ch : h;
    c->stride = ((c->width * bpp) + 15) & ~15;
    c->length = c->stride * c->height;
    c->h_shift = k ? hs : 0;
    c->v_shift = k ? vs : 0;
    total += c->length;
  }
  f->regions[0] = domain->alloc (total);
  if (!f->regions[0]) {
    fprintf (stderr, "domain->alloc (%d) failed: %s\n", total, schro_hip_last_error ());
    exit (1);
  }
  char *p = (char *) f->regions[0];
  for (int k = 0; k < 3; k++) {
    f->components[k].data = p;
    p += f->components[k].length;
  }
  return f;
}

static void
domain_frame_free (SchroHipFrame * f)
{
  f->domain->free (f->regions[0], f->components[0].length + f->components[1].length + f->components[2].length);
  free (f);
}

int
main (int argc, char **argv)
{
  if (argc != 2) {
    fprintf (stderr, "usage: %s DIR\n", argv[0]);
    return 2;
  }
  const char *dir = argv[1];
  char path[1024];
  snprintf (path, sizeof (path), "%s/params.txt", dir);
  FILE *pf = fopen (path, "r");
  int w, h, hs, vs;
  SchroHipParams params;
  memset (&params, 0, sizeof (params));
  if (!pf || fscanf (pf, "%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d", &w, &h, &hs, &vs,
          &params.iwt_luma_width, &params.iwt_luma_height, &params.iwt_chroma_width, &params.iwt_chroma_height,
          &params.transform_depth, &params.wavelet_filter_index, &params.mv_precision, &params.xblen_luma,
          &params.yblen_luma, &params.xbsep_luma, &params.ybsep_luma, &params.x_num_blocks,
          &params.y_num_blocks) != 17) {
    fprintf (stderr, "bad %s\n", path);
    return 2;
  }
  fclose (pf);
  params.num_refs = 2;
  params.picture_weight_bits = 1;
  params.picture_weight_1 = 1;
  params.picture_weight_2 = 1;
  const int cw = (w + (1 << hs) - 1) >> hs, ch = (h + (1 << vs) - 1) >> vs;
  const int fmt8 = SCHRO_HIP_FORMAT_DEPTH_U8 | hs | (vs << 1), fmt16 = SCHRO_HIP_FORMAT_DEPTH_S16 | hs | (vs << 1);
  const size_t pic_bytes = (size_t) w * h + 2 * (size_t) cw * ch;
  const size_t iwt_samples = (size_t) params.iwt_luma_width * params.iwt_luma_height
      + 2 * (size_t) params.iwt_chroma_width * params.iwt_chroma_height;

  /* schro_cuda_init (); decoder->cuda_domain = schro_memory_domain_new_cuda () (schrodecoder.c:167-169) */
  schro_hip_init ();
  SchroHipMemoryDomain *domain = schro_memory_domain_new_hip (0);
  if (!domain) {
    fprintf (stderr, "no HIP domain: %s\n", schro_hip_last_error ());
    return 1;
  }
  SchroHipContext *ctx = schro_hip_domain_context (domain);
  /* the table the reference's slot cache calls (schrodomain.c:58-137) */
  if (!(domain->flags & SCHRO_MEMORY_DOMAIN_HIP))
    return 1;
  void *blk = domain->alloc (1 << 20);
  if (!blk) {
    fprintf (stderr, "domain->alloc failed: %s\n", schro_hip_last_error ());
    return 1;
  }
  CHECK (schro_hip_memset (ctx, blk, 0x11, 1 << 20));
  CHECK (schro_hip_synchronize (ctx));
  domain->free (blk, 1 << 20);

  /* picture->transform_frame: host, iwt-padded, coefficients in the in-place sub-band layout */
  SchroHipFrame transform_frame;
  host_frame (&transform_frame, fmt16, 2, params.iwt_luma_width, params.iwt_luma_height, params.iwt_chroma_width,
      params.iwt_chroma_height, read_file (dir, "coeffs.bin", 2 * iwt_samples));

  /* x_wavelet_transform: picture->frame in the device domain, inverse transform into it */
  SchroHipFrame *frame = domain_frame (domain, fmt16, 2, params.iwt_luma_width, params.iwt_luma_height);
  CHECK (schro_frame_inverse_iwt_transform_hip (frame, &transform_frame, &params));

  /* the two reference pictures (device), x_upsample when mv_precision > 0 */
  SchroHipFrame *ref[2], *src[2];
  for (int r = 0; r < 2; r++) {
    SchroHipFrame hostref;
    host_frame (&hostref, fmt8, 1, w, h, cw, ch, read_file (dir, r ? "ref1.bin" : "ref0.bin", pic_bytes));
    ref[r] = domain_frame (domain, fmt8, 1, w, h);
    CHECK (schro_frame_to_hip (ref[r], &hostref));
    src[r] = ref[r];
    if (params.mv_precision > 0) {
      /* (the half-pel planes have this library's tiled layout: its own allocator) */
      src[r] = schro_hip_frame_new_and_alloc (ctx, fmt8, w, h, 1);
      if (!src[r])
        return 1;
      src[r]->virt_frame1 = ref[r];
      CHECK (schro_upsampled_hipframe_upsample_inplace (src[r]));     /* schrogpuframe.h:29's one-argument form */
    }
    free (hostref.components[0].data);
  }

  /* x_render_motion: schro_motion_render (motion, mc_tmp_frame, frame, TRUE, ref_output_frame) */
  SchroHipMotion motion;
  memset (&motion, 0, sizeof (motion));
  motion.src1 = src[0];
  motion.src2 = src[1];
  motion.motion_vectors = read_file (dir, "mvs.bin", (size_t) 20 * params.x_num_blocks * params.y_num_blocks);
  motion.params = &params;
  SchroHipFrame *output = domain_frame (domain, fmt8, 1, w, h);
  CHECK (schro_motion_render_hip (&motion, NULL, frame, 1, output));

  /* x_combine: the output picture (u8 -> u8 copy), then to the host */
  SchroHipFrame *outpic = domain_frame (domain, fmt8, 1, w, h);
  CHECK (schro_hipframe_convert (outpic, output));
  SchroHipFrame hostout, hostres;
  void *out_bytes = malloc (pic_bytes), *res_bytes = malloc (2 * iwt_samples);
  host_frame (&hostout, fmt8, 1, w, h, cw, ch, out_bytes);
  host_frame (&hostres, fmt16, 2, params.iwt_luma_width, params.iwt_luma_height, params.iwt_chroma_width,
      params.iwt_chroma_height, res_bytes);
  CHECK (schro_hipframe_to_cpu (&hostout, outpic));
  CHECK (schro_hipframe_to_cpu (&hostres, frame));
  write_file (dir, "out.bin", out_bytes, pic_bytes);
  write_file (dir, "residual.bin", res_bytes, 2 * iwt_samples);

  for (int r = 0; r < 2; r++) {
    if (src[r] != ref[r])
      schro_hip_frame_unref (src[r]);
    domain_frame_free (ref[r]);
  }
  domain_frame_free (frame);
  domain_frame_free (output);
  domain_frame_free (outpic);
  schro_memory_domain_free_hip (domain);
  printf ("stage_replay: ok\n");
  return 0;
}
