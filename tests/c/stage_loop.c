/* A C caller of the frame layer (include/schro_hip.h) that decodes a SEQUENCE of inter pictures the way a
 * patched schrodecoder.c would, twice:
 *
 *   (i)  the reference's contract (schroasync-pthread.c:320-328): one picture at a time, one stage per call,
 *        every call complete when it returns -- x_wavelet_transform (schrodecoder.c:1855-1886) ->
 *        schro_frame_inverse_iwt_transform_hip with the HOST transform frame, x_upsample (:2120-2141) once per
 *        reference, x_render_motion (:1697-1792) -> schro_motion_render_hip with the HOST vector array,
 *        x_combine (:1888-2118) -> schro_hipframe_to_cpu;
 *   (ii) INTEGRATION.md 3a: the same stage calls with stage completion OFF
 *        (schro_hip_context_set_stage_completion), frames that cross the boundary in the pinned host domain
 *        (schro_memory_domain_new_hip_host), SLOTS pictures in flight, each with its copies and stages in
 *        order on a queue of its own; the host waits only for the download of the picture it hands on;
 *   (iii) r05 -- (ii) with the QUANTISED hand-over: what goes up per picture is not the dense
 *        transform frame but what a patched schro_decoder_decode_subband keeps (schrodecoder.c:3525-3640): the
 *        codeblock records and one blob of the non-zero codeblocks' quantised values, and
 *        schro_hipframe_dequantise fills the device transform frame in front of x_wavelet_transform.
 *
 * The three passes decode the SAME pictures: the coefficient frames of (i) and (ii) are the quantised sets of (iii),
 * dequantised once by the library at set-up and brought down (the test checks them against the oracle's
 * dequantisation of the dumped records and values).
 *
 * Pictures: W x H 4:2:0, 3-level DD(9,7) residual, 12x12/8x8 quarter-pel OBMC from two references that change
 * every 8 pictures (BASELINE config 3's shape; bench.py's workload).  Inputs are synthetic (an LCG, below);
 * with DUMP = 1 the inputs and every decoded picture of both passes go to DIR for tests/test_gpu_c_harness.py
 * to compare with the oracle.  Prints one JSON line.  (Pass (iii) as bench.py's quantised leg makes its codeblocks:
 * up to 8 x 8 per sub-band, 25 - 75 % of the detail bands' codeblocks zero, two-sided geometric values of one byte.)
 *
 *   stage_loop DIR W H NPICTURES DUMP */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include "schro_hip.h"

#define CHECK(e) do { int r_ = (e); if (r_) { fprintf (stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #e, r_, schro_hip_last_error ()); exit (1); } } while (0)
#define NSETS 4                 /* distinct coefficient frames / vector fields, used in turn */
#define GROUP 8                 /* pictures that share a pair of references */
#define DEPTH 3
#ifndef SLOTS
/* pictures in flight in passes (ii) and (iii).  r05: five (three before).  The host hands over picture k - LAG before it enqueues
 * picture k; with three slots the download of picture k - 1 was enqueued only after the wait for picture k - 2's, so the
 * device-to-host engine idled while the host enqueued -- the quantised pass, whose bound is that engine (12.4 MB per 2160p
 * picture against 7 MB up), measured 0.374 / 0.319 / 0.299 ms per picture with 3 / 4 / 5 slots; the dense pass is the
 * upload's 0.585 whatever the slots.  (Marks: 3 per slot, 16 per context.) */
#define SLOTS 5
#endif
#define LAG (SLOTS - 1)        /* the host hands over picture k - LAG before it enqueues picture k */

static uint32_t lcg_state;
static uint32_t
lcg (void)
{
  lcg_state = lcg_state * 1103515245u + 12345u;
  return lcg_state >> 16;
}

static double
now_ms (void)
{
  struct timespec t;
  clock_gettime (CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static void
write_file (const char *dir, const char *name, int n, const void *p, size_t bytes)
{
  char path[1024];
  snprintf (path, sizeof (path), name, dir, n);
  FILE *f = fopen (path, "wb");
  if (!f || fwrite (p, 1, bytes, f) != bytes) {
    fprintf (stderr, "cannot write %s\n", path);
    exit (2);
  }
  fclose (f);
}

/* schro_frame_new_and_alloc (domain, format, width, height) as the REFERENCE does it (schroframe.c:60-191,
 * extension 0): strides ROUND_UP_16 (width * bytes), the three components back to back in ONE block from
 * the domain's alloc table -- device frames from the HIP domain, pinned host frames from the host domain */
static SchroHipFrame *
domain_frame (SchroHipMemoryDomain * domain, int format, int bpp, int w, int h)
{
  SchroHipFrame *f = (SchroHipFrame *) calloc (1, sizeof (SchroHipFrame));
  const int hs = SCHRO_HIP_FORMAT_H_SHIFT (format), vs = SCHRO_HIP_FORMAT_V_SHIFT (format);
  const int cw = (w + (1 << hs) - 1) >> hs, ch = (h + (1 << vs) - 1) >> vs;
  int total = 0;
  f->refcount = 1;
  f->domain = (domain->flags & SCHRO_MEMORY_DOMAIN_HIP) ? domain : NULL;      /* host frames: domain NULL to this library */
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

static size_t
frame_bytes (const SchroHipFrame * f)
{
  return (size_t) f->components[0].length + f->components[1].length + f->components[2].length;
}

static uint32_t
checksum (const SchroHipFrame * f)
{
  uint32_t h = 2166136261u;
  for (int k = 0; k < 3; k++) {
    const SchroHipFrameData *c = &f->components[k];
    for (int y = 0; y < c->height; y++) {
      const uint8_t *row = (const uint8_t *) c->data + (size_t) y * c->stride;
      for (int x = 0; x < c->width; x += 7)     /* (a sample of the columns: the check is the test's, this is a tripwire) */
        h = (h ^ row[x]) * 16777619u;
    }
  }
  return h;
}

int
main (int argc, char **argv)
{
  if (argc != 6) {
    fprintf (stderr, "usage: %s DIR W H NPICTURES DUMP\n", argv[0]);
    return 2;
  }
  const char *dir = argv[1];
  const int w = atoi (argv[2]), h = atoi (argv[3]), npic = atoi (argv[4]), dump = atoi (argv[5]);
  const int fmt8 = SCHRO_HIP_FORMAT_DEPTH_U8 | 3, fmt16 = SCHRO_HIP_FORMAT_DEPTH_S16 | 3;      /* 4:2:0 */
  const int cw = (w + 1) >> 1, ch = (h + 1) >> 1;
  const int ngroups = (npic + GROUP - 1) / GROUP;
  SchroHipParams params;
  memset (&params, 0, sizeof (params));
  /* schro_params_calculate_iwt_sizes (schroparams.c:75-92), schro_params_calculate_mc_sizes (:165-190) */
  params.transform_depth = DEPTH;
  params.wavelet_filter_index = 0;
  params.iwt_luma_width = (w + 7) & ~7;
  params.iwt_luma_height = (h + 7) & ~7;
  params.iwt_chroma_width = (cw + 7) & ~7;
  params.iwt_chroma_height = (ch + 7) & ~7;
  params.num_refs = 2;
  params.mv_precision = 2;
  params.xblen_luma = params.yblen_luma = 12;
  params.xbsep_luma = params.ybsep_luma = 8;
  params.x_num_blocks = 4 * ((w + 4 * 8 - 1) / (4 * 8));
  params.y_num_blocks = 4 * ((h + 4 * 8 - 1) / (4 * 8));
  params.picture_weight_bits = 1;
  params.picture_weight_1 = params.picture_weight_2 = 1;
  const size_t nmv = (size_t) params.x_num_blocks * params.y_num_blocks;

  schro_hip_init ();
  SchroHipMemoryDomain *domain = schro_memory_domain_new_hip (0);
  SchroHipMemoryDomain *host_domain = schro_memory_domain_new_hip_host ();
  if (!domain || !host_domain) {
    fprintf (stderr, "no HIP domain: %s\n", schro_hip_last_error ());
    return 1;
  }
  SchroHipContext *ctx = schro_hip_domain_context (domain);

  /* ---- inputs, in pinned host frames (pass (i) would work from pageable memory as well) ---- */
  SchroHipFrame *h_coeffs[NSETS], *h_refs[2 * 64];
  uint8_t *h_mvs[NSETS];
  /* the quantised form of each coefficient set: per component the codeblock records (geometry from
   * schro_hip_codeblock_layout with the device transform frame's stride), all three components' values in ONE
   * pinned blob */
  SchroHipCodeblock *q_recs[NSETS][3];
  int q_nrec[3];
  uint8_t *q_blob[NSETS];
  size_t q_off[NSETS][3], q_bytes[NSETS][3], q_total[NSETS], q_cap = 0;
  for (int k = 0; k <= DEPTH; k++)
    params.horiz_codeblocks[k] = params.vert_codeblocks[k] = 8;
  lcg_state = 1;
  for (int s = 0; s < NSETS; s++) {
    h_coeffs[s] = domain_frame (host_domain, fmt16, 2, params.iwt_luma_width, params.iwt_luma_height);
    size_t cap = 0;
    for (int k = 0; k < 3; k++)
      cap += (size_t) h_coeffs[s]->components[k].width * h_coeffs[s]->components[k].height * 2 + 256;
    q_blob[s] = (uint8_t *) host_domain->alloc ((int) cap);
    size_t pos = 0;
    for (int k = 0; k < 3; k++) {
      const SchroHipFrameData *c = &h_coeffs[s]->components[k];        /* (the device transform frames have the same strides) */
      const int n = schro_hip_codeblock_layout (c->width, c->height, DEPTH, params.horiz_codeblocks, params.vert_codeblocks,
          c->stride, 2, NULL, 0);
      q_recs[s][k] = (SchroHipCodeblock *) malloc (sizeof (SchroHipCodeblock) * (size_t) n);
      q_nrec[k] = schro_hip_codeblock_layout (c->width, c->height, DEPTH, params.horiz_codeblocks, params.vert_codeblocks,
          c->stride, 2, q_recs[s][k], n);
      pos = (pos + 255) & ~(size_t) 255;
      q_off[s][k] = pos;
      const size_t base = pos;
      int rec = 0;
      for (int index = 0; index < 1 + 3 * DEPTH; index++) {
        const int level = index == 0 ? 0 : (index - 1) / 3;
        const int per = params.horiz_codeblocks[index == 0 ? 0 : level + 1] * params.vert_codeblocks[index == 0 ? 0 : level + 1];
        /* zero codeblocks: none in the LL band, 25 .. 75 % of the detail bands' towards the finest level */
        const uint32_t p_zero = index == 0 ? 0 : (uint32_t) (25 * level + 25 > 75 ? 75 : 25 * level + 25);
        for (int b = 0; b < per; b++, rec++) {
          SchroHipCodeblock *cb = &q_recs[s][k][rec];
          cb->quant_index = (unsigned char) (index == 0 ? 4 : 6 + 3 * level + (b & 3));   /* (several quantisers per sub-band: codeblock_mode_index 1) */
          if (cb->width == 0 || cb->height == 0 || lcg () % 100 < p_zero)
            continue;           /* src_offset stays -1: a zero codeblock */
          const int wide = index == 0;  /* the LL band's values take two bytes */
          pos = (pos + 1) & ~(size_t) 1;
          cb->src_offset = (int) (pos - base);
          cb->src_bytes = wide ? 2 : 1;
          for (int i = 0; i < cb->width * cb->height; i++) {
            /* two-sided geometric: most values 0 or +-1 */
            uint32_t v = lcg ();
            int mag = 0;
            while ((v & 3) == 3 && mag < 30) {
              mag++;
              v >>= 2;
            }
            if (wide)
              mag = (int) (lcg () % 120);
            const int val = (lcg () & 1) ? -mag : mag;
            if (wide) {
              const int16_t v16 = (int16_t) val;
              memcpy (q_blob[s] + pos, &v16, 2);
              pos += 2;
            } else {
              q_blob[s][pos++] = (uint8_t) (int8_t) val;
            }
          }
        }
      }
      q_bytes[s][k] = pos - base;
    }
    q_total[s] = pos;
    if (pos > q_cap)
      q_cap = pos;
    /* SchroMotionVector records (schromotion.h:20-37): flags @0 (pred_mode in bits 0-1); @12 the union of
     * dx[2], dy[2] and dc[3]; modes 5 / 45 / 15 / 35 %, vectors uniform in +-64 quarter pels */
    h_mvs[s] = (uint8_t *) host_domain->alloc ((int) (20 * nmv));
    memset (h_mvs[s], 0, 20 * nmv);
    for (size_t b = 0; b < nmv; b++) {
      const uint32_t r = lcg () % 100;
      const uint32_t mode = r < 5 ? 0 : r < 50 ? 1 : r < 65 ? 2 : 3;
      int16_t v[4];
      for (int k = 0; k < 4; k++)
        v[k] = mode ? (int16_t) ((int) (lcg () % 129) - 64) : (int16_t) ((int) (lcg () % 256) - 128);
      memcpy (h_mvs[s] + 20 * b, &mode, 4);
      memcpy (h_mvs[s] + 20 * b + 12, v, 8);
    }
  }
  if (ngroups > 64)
    return 2;
  for (int r = 0; r < 2 * ngroups; r++) {
    h_refs[r] = domain_frame (host_domain, fmt8, 1, w, h);
    for (int k = 0; k < 3; k++) {
      SchroHipFrameData *c = &h_refs[r]->components[k];
      for (int y = 0; y < c->height; y++) {
        uint8_t *row = (uint8_t *) c->data + (size_t) y * c->stride;
        for (int x = 0; x < c->width; x++)
          row[x] = (uint8_t) ((lcg () & 0x7f) + ((x + y + 16 * r) & 0x7f));     /* noise on a ramp */
      }
    }
  }
  /* ---- device frames: SLOTS pictures' worth + two reference sets ---- */
  SchroHipFrame *d_transform[SLOTS], *d_frame[SLOTS], *d_out[SLOTS], *h_out[SLOTS], *d_ref[2][2], *d_up[2][2];
  void *d_mv[SLOTS];
  for (int s = 0; s < SLOTS; s++) {
    d_transform[s] = domain_frame (domain, fmt16, 2, params.iwt_luma_width, params.iwt_luma_height);
    d_frame[s] = domain_frame (domain, fmt16, 2, params.iwt_luma_width, params.iwt_luma_height);
    d_out[s] = domain_frame (domain, fmt8, 1, w, h);
    h_out[s] = domain_frame (host_domain, fmt8, 1, w, h);
    d_mv[s] = domain->alloc ((int) (20 * nmv));
  }
  for (int g = 0; g < 2; g++)
    for (int r = 0; r < 2; r++) {
      d_ref[g][r] = domain_frame (domain, fmt8, 1, w, h);
      d_up[g][r] = schro_hip_frame_new_and_alloc (ctx, fmt8, w, h, 1);   /* (this library's tiled half-pel layout) */
      if (!d_up[g][r])
        return 1;
    }
  /* ---- set-up: the dense coefficient frames of passes (i) and (ii) ARE the quantised sets, dequantised by the
   * library (the contract form of the call: host values, complete on return) and brought down ---- */
  void *d_vals[SLOTS];
  for (int s = 0; s < SLOTS; s++)
    d_vals[s] = domain->alloc ((int) q_cap);
  for (int s = 0; s < NSETS; s++) {
    SchroHipQuantisedPicture qp;
    memset (&qp, 0, sizeof (qp));
    for (int k = 0; k < 3; k++) {
      qp.codeblocks[k] = q_recs[s][k];
      qp.ncodeblocks[k] = q_nrec[k];
      qp.values[k] = q_blob[s] + q_off[s][k];
      qp.values_bytes[k] = q_bytes[s][k];
    }
    qp.values_on_device = 0;
    CHECK (schro_hipframe_dequantise (d_transform[0], &qp, &params));
    CHECK (schro_hipframe_to_cpu (h_coeffs[s], d_transform[0]));
  }
  if (dump) {
    for (int s = 0; s < NSETS; s++) {
      write_file (dir, "%s/coeffs%d.bin", s, h_coeffs[s]->regions[0], frame_bytes (h_coeffs[s]));
      write_file (dir, "%s/mvs%d.bin", s, h_mvs[s], 20 * nmv);
      write_file (dir, "%s/qblob%d.bin", s, q_blob[s], q_total[s]);
      for (int k = 0; k < 3; k++) {
        char name[64];
        snprintf (name, sizeof (name), "%%s/qrec%%d_%d.bin", k);
        write_file (dir, name, s, q_recs[s][k], sizeof (SchroHipCodeblock) * (size_t) q_nrec[k]);
      }
    }
    for (int r = 0; r < 2 * ngroups; r++)
      write_file (dir, "%s/ref%d.bin", r, h_refs[r]->regions[0], frame_bytes (h_refs[r]));
  }

  uint32_t *sums[3];
  sums[0] = (uint32_t *) calloc (npic, 4);
  sums[1] = (uint32_t *) calloc (npic, 4);
  sums[2] = (uint32_t *) calloc (npic, 4);
  double ms[3];

  /* ---- pass (i): the reference's contract ---- */
  for (int rep = 0; rep < 2; rep++) {           /* (first repetition, untimed: warm-up -- job tables, allocator -- and the checks) */
    CHECK (schro_hip_synchronize (ctx));
    const double t0 = now_ms ();
    for (int k = 0; k < npic; k++) {
      const int g = k / GROUP, set = k % NSETS;
      if (k % GROUP == 0)
        for (int r = 0; r < 2; r++) {
          /* a reference picture's ref_output_frame is on the device already in a decoder; here it comes from the host */
          CHECK (schro_frame_to_hip (d_ref[g & 1][r], h_refs[2 * g + r]));
          d_up[g & 1][r]->upsample_done = 0;
          CHECK (schro_upsampled_hipframe_upsample (d_up[g & 1][r], d_ref[g & 1][r]));        /* x_upsample */
        }
      CHECK (schro_frame_inverse_iwt_transform_hip (d_frame[0], h_coeffs[set], &params));      /* x_wavelet_transform */
      SchroHipMotion motion;
      memset (&motion, 0, sizeof (motion));
      motion.src1 = d_up[g & 1][0];
      motion.src2 = d_up[g & 1][1];
      motion.motion_vectors = h_mvs[set];
      motion.params = &params;
      CHECK (schro_motion_render_hip (&motion, NULL, d_frame[0], 1, d_out[0]));                /* x_render_motion */
      CHECK (schro_hipframe_to_cpu (h_out[0], d_out[0]));                                      /* x_combine */
      if (rep == 0) {
        sums[0][k] = checksum (h_out[0]);
        if (dump)
          write_file (dir, "%s/out_contract%d.bin", k, h_out[0]->regions[0], frame_bytes (h_out[0]));
      }
    }
    ms[0] = (now_ms () - t0) / npic;
  }

  /* ---- pass (ii): pictures in flight, nothing waits but the hand-over of a finished picture ----
   * Copies up on the H2D queue, the stage calls on queue 0, copies down on the D2H queue, marks between them
   * (INTEGRATION 3a).  The order of the calls matters on this runtime: a hipMemcpyAsync whose queue waits for
   * an event that has NOT fired yet blocks the calling thread until it has, and the copy engines
   * only overlap when the copies sit on queues of their own.  So the host does the one wait a decoder has
   * anyway -- it hands over picture k - LAG (LAG = SLOTS - 1) -- before it enqueues anything of picture k: the frames picture k's
   * upload overwrites were last read by picture k - SLOTS, and the download of picture k - 1 is enqueued once
   * its stage calls are done (a wait of the host, not of the D2H queue).  Measured here: 0.70 ms per picture
   * with a picture's copies and stage calls in order on one queue (three queues in flight); one run in four
   * at that and the others at 1.04 with the reference uploads behind a wait that joined the queues. */
  enum { UP = 0, DONE = SLOTS, DOWN = 2 * SLOTS };      /* marks: + slot (16 marks: SLOTS <= 5) */
  CHECK (schro_hip_context_set_stage_completion (ctx, 0));
  /* pass (iii) is the same loop with the quantised hand-over: the blob of the picture's quantised values goes up instead of
   * the dense transform frame (16 - 20 % of its bytes here), and schro_hipframe_dequantise -- with the values where the
   * copy queue put them -- stands in front of the transform.  The codeblock records are host arrays: the call copies them
   * into the context's pinned table ring, from where a kernel brings them over (no copy-engine call of its own). */
  double host_ms[3][4];         /* [pass][waited, uploads, stages, download]: where the host's time goes in the timed repetition */
  memset (host_ms, 0, sizeof (host_ms));
  for (int pass = 1; pass <= 2; pass++) {
  const int quant = pass == 2;
  double waited = 0, t_up = 0, t_stage = 0, t_down = 0;
  for (int rep = 0; rep < 2; rep++) {
    CHECK (schro_hip_synchronize (ctx));
    waited = t_up = t_stage = t_down = 0;
    const double t0 = now_ms ();
    for (int k = 0; k < npic + LAG; k++) {
      const int s = k % SLOTS;
      if (k >= LAG) {
        /* picture k - LAG leaves: the host waits for ITS download only */
        const int s2 = (k - LAG) % SLOTS;
        const double tw = now_ms ();
        CHECK (schro_hip_queue_mark_synchronize (ctx, DOWN + s2));
        waited += now_ms () - tw;
        if (rep == 0) {
          sums[pass][k - LAG] = checksum (h_out[s2]);
          if (dump)
            write_file (dir, quant ? "%s/out_quantised%d.bin" : "%s/out_pipelined%d.bin", k - LAG, h_out[s2]->regions[0], frame_bytes (h_out[s2]));
        }
      }
      if (k < npic) {
        const int g = k / GROUP, set = k % NSETS;
        const double ta = now_ms ();
        /* the transform frame and the vectors up (pinned host memory); a group's first picture also brings the new
         * references (a decoder has them on the device already) */
        CHECK (schro_hip_context_select_queue (ctx, SCHRO_HIP_QUEUE_H2D));
        if (quant)
          CHECK (schro_hip_upload_2d_async (ctx, d_vals[s], (int) q_total[set], q_blob[set], (int) q_total[set], (int) q_total[set], 1));
        else
          CHECK (schro_frame_to_hip_async (d_transform[s], h_coeffs[set]));
        CHECK (schro_hip_upload_2d_async (ctx, d_mv[s], (int) (20 * nmv), h_mvs[set], (int) (20 * nmv), (int) (20 * nmv), 1));
        if (k % GROUP == 0)
          for (int r = 0; r < 2; r++)
            CHECK (schro_frame_to_hip_async (d_ref[g & 1][r], h_refs[2 * g + r]));
        CHECK (schro_hip_queue_mark (ctx, UP + s));
        const double tb = now_ms ();
        t_up += tb - ta;
        CHECK (schro_hip_context_select_queue (ctx, 0));
        CHECK (schro_hip_queue_wait_mark (ctx, UP + s));         /* (kernels behind a wait cost the host nothing) */
        if (k % GROUP == 0)
          for (int r = 0; r < 2; r++) {
            d_up[g & 1][r]->upsample_done = 0;
            CHECK (schro_upsampled_hipframe_upsample (d_up[g & 1][r], d_ref[g & 1][r]));     /* x_upsample */
          }
        if (quant) {
          /* what schro_decoder_decode_subband left behind for this picture */
          SchroHipQuantisedPicture qp;
          memset (&qp, 0, sizeof (qp));
          for (int c = 0; c < 3; c++) {
            qp.codeblocks[c] = q_recs[set][c];
            qp.ncodeblocks[c] = q_nrec[c];
            qp.values[c] = (const char *) d_vals[s] + q_off[set][c];
            qp.values_bytes[c] = q_bytes[set][c];
          }
          qp.values_on_device = 1;
          CHECK (schro_hipframe_dequantise (d_transform[s], &qp, &params));
        }
        CHECK (schro_frame_inverse_iwt_transform_hip (d_frame[s], d_transform[s], &params));  /* x_wavelet_transform */
        SchroHipMotion motion;
        memset (&motion, 0, sizeof (motion));
        motion.src1 = d_up[g & 1][0];
        motion.src2 = d_up[g & 1][1];
        motion.motion_vectors = d_mv[s];        /* already on the device */
        motion.params = &params;
        CHECK (schro_motion_render_hip (&motion, NULL, d_frame[s], 1, d_out[s]));             /* x_render_motion */
        CHECK (schro_hip_queue_mark (ctx, DONE + s));
        t_stage += now_ms () - tb;
      }
      if (k >= 1 && k - 1 < npic) {
        /* x_combine of picture k - 1: its stage calls have had the time of picture k's enqueue; the host makes sure,
         * so that the copy call finds its event fired */
        const int s1 = (k - 1) % SLOTS;
        const double tw = now_ms ();
        CHECK (schro_hip_queue_mark_synchronize (ctx, DONE + s1));
        const double tc = now_ms ();
        waited += tc - tw;
        CHECK (schro_hip_context_select_queue (ctx, SCHRO_HIP_QUEUE_D2H));
        CHECK (schro_hipframe_to_cpu_async (h_out[s1], d_out[s1]));
        CHECK (schro_hip_queue_mark (ctx, DOWN + s1));
        t_down += now_ms () - tc;
      }
    }
    CHECK (schro_hip_context_select_queue (ctx, 0));
    CHECK (schro_hip_synchronize (ctx));
    ms[pass] = (now_ms () - t0) / npic;
  }
  host_ms[pass][0] = waited / npic;
  host_ms[pass][1] = t_up / npic;
  host_ms[pass][2] = t_stage / npic;
  host_ms[pass][3] = t_down / npic;
  }
  int same = 1;
  size_t dense_bytes = frame_bytes (h_coeffs[0]), q_sum = 0;
  for (int s = 0; s < NSETS; s++) {
    q_sum += q_total[s];
    for (int k = 0; k < 3; k++)
      q_sum += sizeof (SchroHipCodeblock) * (size_t) q_nrec[k];
  }
  for (int k = 0; k < npic; k++)
    same &= sums[0][k] == sums[1][k] && sums[0][k] == sums[2][k];
  printf ("{\"width\": %d, \"height\": %d, \"pictures\": %d, \"contract_ms_per_picture\": %.4f, \"pipelined_ms_per_picture\": %.4f, "
      "\"quantised_pipelined_ms_per_picture\": %.4f, "
      "\"contract_Mpix_per_s\": %.1f, \"pipelined_Mpix_per_s\": %.1f, \"quantised_pipelined_Mpix_per_s\": %.1f, "
      "\"pictures_in_flight\": %d, \"passes_agree\": %s, "
      "\"quantised_share_of_dense_coefficient_bytes\": %.3f, "
      "\"pipelined_host_ms_per_picture\": {\"waiting_for_a_finished_picture\": %.4f, \"enqueue_uploads\": %.4f, "
      "\"enqueue_stages\": %.4f, \"enqueue_download\": %.4f}, "
      "\"quantised_pipelined_host_ms_per_picture\": {\"waiting_for_a_finished_picture\": %.4f, \"enqueue_uploads\": %.4f, "
      "\"enqueue_stages\": %.4f, \"enqueue_download\": %.4f}}\n", w, h, npic,
      ms[0], ms[1], ms[2], (double) w * h / ms[0] / 1e3, (double) w * h / ms[1] / 1e3, (double) w * h / ms[2] / 1e3, SLOTS,
      same ? "true" : "false", (double) q_sum / NSETS / (double) dense_bytes,
      host_ms[1][0], host_ms[1][1], host_ms[1][2], host_ms[1][3], host_ms[2][0], host_ms[2][1], host_ms[2][2], host_ms[2][3]);
  return same ? 0 : 3;
}
