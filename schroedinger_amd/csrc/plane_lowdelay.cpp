// plane_lowdelay.cpp -- plane layer: VC-2 low-delay slices and DC prediction (lowdelay.hip), core-syntax dequantisation,
// its plans and the codeblock layout (dequant.hip).

#include "schro_hip_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

using namespace schro;

extern "C" {

// ---- VC-2 low-delay transform data (lowdelay.hip) ---------------------------------

int
schro_hip_lowdelay_arith (const SchroHipLowDelayParams * p, int bytes_per_sample)
{
  SCHRO_HIP_REQUIRE (p && (bytes_per_sample == 2 || bytes_per_sample == 4) && p->n_horiz_slices > 0
      && p->n_vert_slices > 0 && p->transform_depth >= 0 && p->transform_depth <= 6, "lowdelay_arith: bad arguments");
  if (bytes_per_sample == 4)
    return SCHRO_HIP_LOWDELAY_S32;
  // schrolowdelay.c:751-760
  if ((p->iwt_chroma_width >> p->transform_depth) % p->n_horiz_slices == 0
      && (p->iwt_chroma_height >> p->transform_depth) % p->n_vert_slices == 0)
    return SCHRO_HIP_LOWDELAY_FAST16;
  return SCHRO_HIP_LOWDELAY_SLOW16;
}

int
schro_hip_dc_predict_batch (SchroHipContext * ctx, const SchroHipDcPlane * planes, int nplanes,
    int bytes_per_sample)
{
  SCHRO_HIP_REQUIRE (ctx && planes && nplanes > 0 && nplanes <= 3 * kMaxJobs, "dc_predict_batch: bad arguments");
  SCHRO_HIP_REQUIRE (bytes_per_sample == 2 || bytes_per_sample == 4, "dc_predict_batch: bytes_per_sample must be 2 or 4");
  (void) hipSetDevice (ctx->device);
  std::vector < DcJob > jobs (nplanes);
  int max_rows = 1, max_w = 1;
  for (int p = 0; p < nplanes; p++) {
    const SchroHipDcPlane & pl = planes[p];
    SCHRO_HIP_REQUIRE (pl.data && pl.width > 0 && pl.height > 0 && pl.stride >= pl.width * bytes_per_sample
        && pl.stride % bytes_per_sample == 0 && (uintptr_t) pl.data % bytes_per_sample == 0,
        "dc_predict_batch: plane %d invalid", p);
    jobs[p].data = pl.data;
    jobs[p].stride = pl.stride;
    jobs[p].w = pl.width;
    jobs[p].h = pl.height;
    jobs[p].pad = 0;
    max_rows = std::max (max_rows, pl.height);
    max_w = std::max (max_w, pl.width);
  }
  void *d_jobs;
  int r = push_args (ctx, jobs.data (), sizeof (DcJob) * nplanes, &d_jobs);
  if (r)
    return r;
  unsigned long long *edge = nullptr;
  int edge_pitch = 0;
  uint32_t epoch = 0;
  if (dc_skew_ok (jobs.data (), nplanes, bytes_per_sample)) {
    r = dc_edge_for (ctx, nplanes, max_rows, max_w, &edge, &edge_pitch, &epoch);
    if (r)
      return r;
  }
  ProfileScope ps (ctx, SCHRO_HIP_KERNEL_DC_PREDICT);
  return launch_dc_predict (ctx->stream, (const DcJob *) d_jobs, nplanes, max_rows, bytes_per_sample, edge, edge_pitch,
      epoch, ctx->dc_gave_up);
}

int
schro_hip_dequant_batch (SchroHipContext * ctx, const SchroHipDequantPlane * planes, int nplanes, int bytes_per_sample,
    int arith)
{
  SCHRO_HIP_REQUIRE (ctx && planes && nplanes > 0, "dequant_batch: bad arguments");
  SCHRO_HIP_REQUIRE (bytes_per_sample == 2 || bytes_per_sample == 4, "dequant_batch: bytes_per_sample must be 2 or 4");
  SCHRO_HIP_REQUIRE (arith == 0 || (arith == 1 && bytes_per_sample == 2),
      "dequant_batch: the 16-bit arithmetic belongs to s16 frames");
  (void) hipSetDevice (ctx->device);
  int tw, th;
  dequant_tile_geometry (&tw, &th);
  // one launch per 2^18 codeblocks (find_dequant_job's three probes): the table of a whole batch of
  // pictures goes up as one copy (r03; through the 64 KB table slots it was a launch per 1365
  // codeblocks -- eleven launches of 20 us for 8 x 2160p)
  // (SCHRO_HIP_DEQUANT_PER_LAUNCH=1300: tables that fit the slots again, for A/B runs)
  const char *env = SCHRO_ENV ("SCHRO_HIP_DEQUANT_PER_LAUNCH");
  const size_t kPerLaunch = env && atoi (env) > 0 ? std::min ((size_t) atoi (env), (size_t) 1 << 18) : (size_t) 1 << 18;
  std::vector < DequantJob > jobs;
  std::vector < char > table;
  int tile_base = 0;
  auto flush = [&] () -> int {
    if (jobs.empty ())
      return 0;
    // behind the jobs: their first tiles, every 64th and every 4096th of them, each as a dense array
    // (find_dequant_job's probes read 64 neighbouring words instead of 64 job records)
    const size_t n = jobs.size (), n64 = (n + 63) / 64, n4096 = (n + 4095) / 4096;
    const size_t bytes = sizeof (DequantJob) * n + sizeof (int) * (n + n64 + n4096);
    table.resize (bytes);
    memcpy (table.data (), jobs.data (), sizeof (DequantJob) * n);
    int *index = (int *) (table.data () + sizeof (DequantJob) * n);
    for (size_t k = 0; k < n; k++)
      index[k] = jobs[k].tile_base;
    for (size_t k = 0; k < n64; k++)
      index[n + k] = jobs[64 * k].tile_base;
    for (size_t k = 0; k < n4096; k++)
      index[n + n64 + k] = jobs[4096 * k].tile_base;
    void *d_jobs;
    int r = bytes <= SchroHipContext::kArgSlotBytes ? push_args (ctx, table.data (), bytes, &d_jobs)
        : push_big_table (ctx, table.data (), bytes, &d_jobs);
    if (r)
      return r;
    {
      ProfileScope ps (ctx, SCHRO_HIP_KERNEL_DEQUANT);
      r = launch_dequant (ctx->stream, (const DequantJob *) d_jobs, (int) jobs.size (), tile_base, bytes_per_sample, arith);
    }
    jobs.clear ();
    tile_base = 0;
    return r;
  };
  for (int p = 0; p < nplanes; p++) {
    const SchroHipDequantPlane & pl = planes[p];
    SCHRO_HIP_REQUIRE (pl.dst && pl.codeblocks && pl.ncodeblocks > 0 && (uintptr_t) pl.dst % bytes_per_sample == 0,
        "dequant_batch: plane %d invalid", p);
    for (int c = 0; c < pl.ncodeblocks; c++) {
      const SchroHipCodeblock & cb = pl.codeblocks[c];
      if (cb.width == 0 || cb.height == 0)
        continue;                 // (a sub-band narrower than its codeblock count: schrodecoder.c:3572-3588)
      SCHRO_HIP_REQUIRE (cb.width > 0 && cb.height > 0 && cb.dst_offset >= 0 && cb.dst_offset % bytes_per_sample == 0
          && cb.dst_stride % bytes_per_sample == 0 && cb.dst_stride >= cb.width * bytes_per_sample,
          "dequant_batch: plane %d codeblock %d: bad geometry", p, c);
      SCHRO_HIP_REQUIRE (cb.src_offset < 0 || (pl.values && (cb.src_bytes == 1 || cb.src_bytes == 2 || cb.src_bytes == 4)
              && cb.src_offset % cb.src_bytes == 0),
          "dequant_batch: plane %d codeblock %d: values must be 1, 2 or 4 bytes each and aligned", p, c);
      SCHRO_HIP_REQUIRE (cb.quant_index <= 60, "dequant_batch: plane %d codeblock %d: quant_index %d", p, c, cb.quant_index);
      DequantJob j;
      memset (&j, 0, sizeof (j));
      j.dst = (char *) pl.dst + cb.dst_offset;
      j.src = cb.src_offset < 0 ? nullptr : (const char *) pl.values + cb.src_offset;
      j.dst_stride = cb.dst_stride;
      j.w = cb.width;
      j.h = cb.height;
      j.src_bytes = cb.src_bytes;
      dequant_tables (cb.quant_index, pl.is_intra, &j.factor, &j.offset);
      j.tiles_x = div_up (cb.width, tw);
      j.tile_base = tile_base;
      tile_base += j.tiles_x * div_up (cb.height, th);
      jobs.push_back (j);
      if (jobs.size () == kPerLaunch) {
        int r = flush ();
        if (r)
          return r;
      }
    }
  }
  return flush ();
}

// ---- r04: dequantisation plans -- the host cost of a repeated picture geometry is O (planes) ----------
// schro_hip_dequant_batch turns every codeblock record into a 48-byte job on the host, every call: 15 k records
// per 8 x 2160p, 1.3 ms of a 2.3 ms PCIe-inclusive step (bench.py pcie_inclusive_quantised, r03).  But what a
// decoder knows per picture GEOMETRY (schro_hip_codeblock_layout: rectangles, pitches -- and so the tiles of
// the launch and which codeblock owns which) never changes; what its entropy decoder produces per PICTURE
// (src_offset / src_bytes / quant_index of each record, the values) the kernel can read for itself.  A plan
// is the first part, resident on the device; a run uploads the records as they are (24 bytes each, one copy
// through a pinned mirror) and a line per plane.
struct SchroHipDequantPlan {
  SchroHipContext *ctx;
  int bpp, arith;
  int njobs, total_tiles;
  size_t total_recs;
  std::vector < int >ncb;       // records per plane
  std::vector < SchroHipCodeblock > geo;        // the plan's copy of every record's geometry (checked per run)
  void *d_geo;                  // DequantGeo[njobs] + the three first-tile index arrays
};

SchroHipDequantPlan *
schro_hip_dequant_plan_new (SchroHipContext * ctx, const SchroHipDequantPlane * planes, int nplanes, int bytes_per_sample, int arith)
{
  if (!ctx || !planes || nplanes <= 0 || nplanes > 4096 || (bytes_per_sample != 2 && bytes_per_sample != 4)
      || !(arith == 0 || (arith == 1 && bytes_per_sample == 2))) {
    set_error (SCHRO_HIP_EINVAL, "dequant_plan_new: bad arguments");
    return nullptr;
  }
  (void) hipSetDevice (ctx->device);
  int tw, th;
  dequant_tile_geometry (&tw, &th);
  SchroHipDequantPlan *plan = new SchroHipDequantPlan ();
  plan->ctx = ctx;
  plan->bpp = bytes_per_sample;
  plan->arith = arith;
  plan->d_geo = nullptr;
  std::vector < DequantGeo > jobs;
  int tile_base = 0;
  size_t rec = 0;
  for (int p = 0; p < nplanes; p++) {
    const SchroHipDequantPlane & pl = planes[p];
    if (!pl.codeblocks || pl.ncodeblocks <= 0) {
      set_error (SCHRO_HIP_EINVAL, "dequant_plan_new: plane %d has no codeblock records", p);
      delete plan;
      return nullptr;
    }
    plan->ncb.push_back (pl.ncodeblocks);
    for (int c = 0; c < pl.ncodeblocks; c++, rec++) {
      const SchroHipCodeblock & cb = pl.codeblocks[c];
      plan->geo.push_back (cb);
      if (cb.width == 0 || cb.height == 0)
        continue;               // (a sub-band narrower than its codeblock count: schrodecoder.c:3572-3588)
      if (cb.width < 0 || cb.height < 0 || cb.dst_offset < 0 || cb.dst_offset % bytes_per_sample
          || cb.dst_stride % bytes_per_sample || cb.dst_stride < cb.width * bytes_per_sample) {
        set_error (SCHRO_HIP_EINVAL, "dequant_plan_new: plane %d codeblock %d: bad geometry", p, c);
        delete plan;
        return nullptr;
      }
      DequantGeo g;
      g.dst_offset = cb.dst_offset;
      g.dst_stride = cb.dst_stride;
      g.w = cb.width;
      g.h = cb.height;
      g.tiles_x = div_up (cb.width, tw);
      g.tile_base = tile_base;
      g.plane = p;
      g.rec = (int) rec;
      tile_base += g.tiles_x * div_up (cb.height, th);
      jobs.push_back (g);
    }
  }
  plan->total_recs = rec;
  plan->njobs = (int) jobs.size ();
  plan->total_tiles = tile_base;
  if (jobs.empty () || jobs.size () > ((size_t) 1 << 18)) {
    set_error (SCHRO_HIP_EINVAL, "dequant_plan_new: %zu codeblocks (1 .. 2^18 per plan)", jobs.size ());
    delete plan;
    return nullptr;
  }
  // behind the jobs: their first tiles, every 64th and every 4096th of them (find_dequant_job's probes)
  const size_t n = jobs.size (), n64 = (n + 63) / 64, n4096 = (n + 4095) / 4096;
  const size_t bytes = sizeof (DequantGeo) * n + sizeof (int) * (n + n64 + n4096);
  std::vector < char >table (bytes);
  memcpy (table.data (), jobs.data (), sizeof (DequantGeo) * n);
  int *index = (int *) (table.data () + sizeof (DequantGeo) * n);
  for (size_t k = 0; k < n; k++)
    index[k] = jobs[k].tile_base;
  for (size_t k = 0; k < n64; k++)
    index[n + k] = jobs[64 * k].tile_base;
  for (size_t k = 0; k < n4096; k++)
    index[n + n64 + k] = jobs[4096 * k].tile_base;
  if (hipMalloc (&plan->d_geo, bytes) != hipSuccess
      || hipMemcpy (plan->d_geo, table.data (), bytes, hipMemcpyHostToDevice) != hipSuccess) {
    set_error (SCHRO_HIP_ENOMEM, "dequant_plan_new: %zu bytes of plan", bytes);
    if (plan->d_geo)
      (void) hipFree (plan->d_geo);
    delete plan;
    return nullptr;
  }
  return plan;
}

// whether `planes` are pictures of the plan's geometry (frame layer: one plan per context, rebuilt when this says no)
bool
schro_hip_dequant_plan_matches (const SchroHipDequantPlan * plan, const SchroHipDequantPlane * planes, int nplanes, int bpp, int arith)
{
  if (!plan || plan->bpp != bpp || plan->arith != arith || nplanes != (int) plan->ncb.size ())
    return false;
  size_t rec = 0;
  for (int p = 0; p < nplanes; p++) {
    if (planes[p].ncodeblocks != plan->ncb[p])
      return false;
    const SchroHipCodeblock *g = plan->geo.data () + rec, *c = planes[p].codeblocks;
    unsigned bad = 0;
    for (int k = 0; k < planes[p].ncodeblocks; k++)
      bad |= (unsigned) (c[k].dst_offset ^ g[k].dst_offset) | (unsigned) (c[k].dst_stride ^ g[k].dst_stride)
          | (unsigned) (c[k].width ^ g[k].width) | (unsigned) (c[k].height ^ g[k].height);
    if (bad)
      return false;
    rec += (size_t) planes[p].ncodeblocks;
  }
  return true;
}

void
schro_hip_dequant_plan_free (SchroHipDequantPlan * plan)
{
  if (!plan)
    return;
  (void) hipSetDevice (plan->ctx->device);
  for (int q = 0; q < SchroHipContext::kQueues; q++)     // launches that still read the plan
    (void) hipStreamSynchronize (plan->ctx->streams[q]);
  (void) hipFree (plan->d_geo);
  delete plan;
}

// planes: as given to schro_hip_dequant_plan_new, with this picture batch's dst / values pointers and
// records -- the records' src_offset, src_bytes and quant_index are read (by the device); their geometry
// must be the plan's
int
schro_hip_dequant_plan_run (SchroHipDequantPlan * plan, const SchroHipDequantPlane * planes, int nplanes)
{
  SCHRO_HIP_REQUIRE (plan && planes && nplanes == (int) plan->ncb.size (), "dequant_plan_run: bad arguments");
  SchroHipContext *ctx = plan->ctx;
  (void) hipSetDevice (ctx->device);
  const size_t rec_bytes = (plan->total_recs * sizeof (SchroHipCodeblock) + 15) & ~(size_t) 15;
  const size_t bytes = rec_bytes + sizeof (DequantPlaneDyn) * (size_t) nplanes;
  void *h, *d;
  int r = big_table_begin (ctx, bytes, &h, &d);
  if (r)
    return r;
  SchroHipCodeblock *recs = (SchroHipCodeblock *) h;
  DequantPlaneDyn *dyn = (DequantPlaneDyn *) ((char *) h + rec_bytes);
  size_t rec = 0;
  for (int p = 0; p < nplanes; p++) {
    const SchroHipDequantPlane & pl = planes[p];
    SCHRO_HIP_REQUIRE (pl.dst && pl.codeblocks && pl.ncodeblocks == plan->ncb[p] && (uintptr_t) pl.dst % plan->bpp == 0,
        "dequant_plan_run: plane %d does not match the plan", p);
    memcpy (recs + rec, pl.codeblocks, sizeof (SchroHipCodeblock) * (size_t) pl.ncodeblocks);
    // what the device will trust: geometry = the plan's, values addressable, quantiser in the tables
    const SchroHipCodeblock *g = plan->geo.data () + rec, *c = pl.codeblocks;
    unsigned bad = 0;
    for (int k = 0; k < pl.ncodeblocks; k++) {
      bad |= (unsigned) (c[k].dst_offset ^ g[k].dst_offset) | (unsigned) (c[k].dst_stride ^ g[k].dst_stride)
          | (unsigned) (c[k].width ^ g[k].width) | (unsigned) (c[k].height ^ g[k].height);
      const bool has = c[k].src_offset >= 0 && c[k].width > 0 && c[k].height > 0;
      const unsigned sb = c[k].src_bytes;
      bad |= has && !(pl.values && (sb == 1 || sb == 2 || sb == 4) && c[k].src_offset % (int) sb == 0);
      bad |= c[k].quant_index > 60;
    }
    SCHRO_HIP_REQUIRE (!bad, "dequant_plan_run: plane %d: a record's geometry differs from the plan's, or its values are "
        "not 1 / 2 / 4 bytes each and aligned, or its quant_index is above 60", p);
    dyn[p].dst = pl.dst;
    dyn[p].values = pl.values;
    dyn[p].is_intra = pl.is_intra;
    dyn[p].pad = 0;
    rec += (size_t) pl.ncodeblocks;
  }
  r = big_table_commit (ctx, bytes);
  if (r)
    return r;
  ProfileScope ps (ctx, SCHRO_HIP_KERNEL_DEQUANT);
  return launch_dequant_plan (ctx->stream, (const DequantGeo *) plan->d_geo, plan->njobs, plan->total_tiles,
      (const SchroHipCodeblock *) d, (const DequantPlaneDyn *) ((const char *) d + rec_bytes), plan->bpp, plan->arith);
}

// The codeblock records of one component in the decoder's order: sub-band index 0 .. 3 * depth
// (position by schro_subband_get_position, schroparams.c:355-368; rectangle by
// schro_subband_get_frame_data, :319-352), in each the rows of codeblocks of
// schro_decoder_decode_subband (schrodecoder.c:3558-3577; their counts by
// schro_decoder_setup_codeblocks, :3280-3293).  Geometry only: src_offset -1 (zero codeblock),
// src_bytes 0, quant_index 0 -- what the entropy decoder fills in as it goes.
int
schro_hip_codeblock_layout (int iwt_width, int iwt_height, int transform_depth, const int *horiz_codeblocks,
    const int *vert_codeblocks, int stride, int bytes_per_sample, SchroHipCodeblock * out, int max)
{
  SCHRO_HIP_REQUIRE (iwt_width > 0 && iwt_height > 0 && transform_depth >= 0 && transform_depth <= 6 && horiz_codeblocks
      && vert_codeblocks && stride > 0 && (bytes_per_sample == 2 || bytes_per_sample == 4) && (out || max == 0),
      "codeblock_layout: bad arguments");
  int n = 0;
  for (int index = 0; index < 1 + 3 * transform_depth; index++) {
    const int position = index == 0 ? 0 : (((index - 1) / 3) << 2) | ((index - 1) % 3 + 1);
    const int level = position >> 2;                    // SCHRO_SUBBAND_SHIFT
    const int shift = transform_depth - level;
    const int bw = iwt_width >> shift, bh = iwt_height >> shift;
    const int bstride = stride << shift;
    const int base = ((position & 2) ? bstride >> 1 : 0) + ((position & 1) ? bw * bytes_per_sample : 0);
    const int hc = horiz_codeblocks[position == 0 ? 0 : level + 1], vc = vert_codeblocks[position == 0 ? 0 : level + 1];
    SCHRO_HIP_REQUIRE (hc > 0 && vc > 0, "codeblock_layout: sub-band %d has %d x %d codeblocks", index, hc, vc);
    for (int y = 0; y < vc; y++) {
      const int ymin = (bh * y) / vc, ymax = (bh * (y + 1)) / vc;
      int xmin = 0, acc = 0;
      const int cw = bw / hc, inc = bw - hc * cw;
      for (int x = 0; x < hc; x++) {
        const int x0 = xmin;
        xmin += cw;
        acc += inc;
        if (acc >= hc) {
          acc -= hc;
          xmin++;
        }
        if (n < max) {
          SchroHipCodeblock & cb = out[n];
          cb.dst_offset = base + ymin * bstride + x0 * bytes_per_sample;
          cb.dst_stride = bstride;
          cb.width = xmin - x0;
          cb.height = ymax - ymin;
          cb.src_offset = -1;
          cb.src_bytes = 0;
          cb.quant_index = 0;
          cb.pad[0] = cb.pad[1] = 0;
        }
        n++;
      }
    }
  }
  return n;
}

int
schro_hip_lowdelay_batch (SchroHipContext * ctx, const SchroHipLowDelayPicture * pictures, int npictures,
    const SchroHipLowDelayParams * params, int bytes_per_sample)
{
  SCHRO_HIP_REQUIRE (ctx && pictures && params && npictures > 0 && npictures <= kMaxJobs,
      "lowdelay_batch: bad arguments");
  SCHRO_HIP_REQUIRE (bytes_per_sample == 2 || bytes_per_sample == 4, "lowdelay_batch: bytes_per_sample must be 2 or 4");
  const SchroHipLowDelayParams & lp = *params;
  const int depth = lp.transform_depth;
  SCHRO_HIP_REQUIRE (depth >= 0 && depth <= 6, "lowdelay_batch: transform_depth %d", depth);
  SCHRO_HIP_REQUIRE (lp.iwt_luma_width > 0 && lp.iwt_luma_height > 0 && lp.iwt_chroma_width > 0
      && lp.iwt_chroma_height > 0 && ((lp.iwt_luma_width | lp.iwt_luma_height | lp.iwt_chroma_width
              | lp.iwt_chroma_height) & ((1 << depth) - 1)) == 0,
      "lowdelay_batch: iwt sizes must be positive multiples of 2^depth");
  SCHRO_HIP_REQUIRE (lp.n_horiz_slices > 0 && lp.n_vert_slices > 0
      && (int64_t) lp.n_horiz_slices * lp.n_vert_slices < (1 << 24), "lowdelay_batch: bad slice counts");
  SCHRO_HIP_REQUIRE (lp.slice_bytes_denom > 0 && lp.slice_bytes_num >= lp.slice_bytes_denom,
      "lowdelay_batch: slice_bytes %d / %d", lp.slice_bytes_num, lp.slice_bytes_denom);
  const int arith = schro_hip_lowdelay_arith (params, bytes_per_sample);
  if (arith < 0)
    return arith;
  // schrodecoder.c:2931-2932: the slices of a picture take num * slices / denom bytes
  const int64_t nslices = (int64_t) lp.n_horiz_slices * lp.n_vert_slices;
  const int64_t need = ((int64_t) lp.slice_bytes_num * nslices) / lp.slice_bytes_denom;
  SCHRO_HIP_REQUIRE (need < ((int64_t) 1 << 28), "lowdelay_batch: %lld bytes of slices per picture", (long long) need);
  (void) hipSetDevice (ctx->device);

  SliceParams P;
  memset (&P, 0, sizeof (P));
  P.depth = depth;
  P.iwt_lw = lp.iwt_luma_width;
  P.iwt_lh = lp.iwt_luma_height;
  P.iwt_cw = lp.iwt_chroma_width;
  P.iwt_ch = lp.iwt_chroma_height;
  P.nh = lp.n_horiz_slices;
  P.nv = lp.n_vert_slices;
  P.n_bytes = lp.slice_bytes_num / lp.slice_bytes_denom;
  P.remainder = lp.slice_bytes_num % lp.slice_bytes_denom;
  P.denom = lp.slice_bytes_denom;
  for (int i = 0; i < 1 + 3 * depth; i++)
    P.quant_matrix[i] = lp.quant_matrix[i];

  std::vector < SliceJob > jobs (npictures);
  std::vector < DcJob > dc (3 * (size_t) npictures);
  bool aligned16 = true;
  for (int p = 0; p < npictures; p++) {
    const SchroHipLowDelayPicture & pic = pictures[p];
    SCHRO_HIP_REQUIRE (pic.slices && (int64_t) pic.slices_bytes >= need && pic.slices_bytes < ((size_t) 1 << 28),
        "lowdelay_batch: picture %d: %zu bytes of slices, %lld needed", p, pic.slices_bytes, (long long) need);
    SliceJob & j = jobs[p];
    memset (&j, 0, sizeof (j));
    j.data = pic.slices;
    j.data_bytes = (uint32_t) pic.slices_bytes;
    for (int k = 0; k < 3; k++) {
      const int w = k ? lp.iwt_chroma_width : lp.iwt_luma_width;
      SCHRO_HIP_REQUIRE (pic.comp[k] && pic.stride[k] >= w * bytes_per_sample && pic.stride[k] % bytes_per_sample == 0
          && (uintptr_t) pic.comp[k] % bytes_per_sample == 0, "lowdelay_batch: picture %d component %d invalid", p, k);
      j.comp[k] = pic.comp[k];
      j.stride[k] = pic.stride[k];
      aligned16 = aligned16 && (((uintptr_t) pic.comp[k] | (uintptr_t) pic.stride[k]) & 15) == 0;
      DcJob & d = dc[3 * (size_t) p + k];
      d.data = pic.comp[k];             // the LL band: sub-band 0 of schro_subband_get_frame_data
      d.stride = pic.stride[k] << depth;
      d.w = w >> depth;
      d.h = (k ? lp.iwt_chroma_height : lp.iwt_luma_height) >> depth;
      d.pad = 0;
    }
  }
  void *d_jobs;
  int r = push_args (ctx, jobs.data (), sizeof (SliceJob) * npictures, &d_jobs);
  if (r)
    return r;
  {
    ProfileScope ps (ctx, SCHRO_HIP_KERNEL_SLICES);
    r = launch_slices (ctx->stream, (const SliceJob *) d_jobs, npictures, P, bytes_per_sample, arith, aligned16);
    if (r)
      return r;
  }
  void *d_dc;
  r = push_args (ctx, dc.data (), sizeof (DcJob) * dc.size (), &d_dc);
  if (r)
    return r;
  unsigned long long *edge = nullptr;
  int edge_pitch = 0;
  uint32_t epoch = 0;
  if (dc_skew_ok (dc.data (), (int) dc.size (), bytes_per_sample)) {
    r = dc_edge_for (ctx, (int) dc.size (), lp.iwt_luma_height >> depth, lp.iwt_luma_width >> depth, &edge, &edge_pitch,
        &epoch);
    if (r)
      return r;
  }
  ProfileScope ps (ctx, SCHRO_HIP_KERNEL_DC_PREDICT);
  return launch_dc_predict (ctx->stream, (const DcJob *) d_dc, (int) dc.size (), lp.iwt_luma_height >> depth,
      bytes_per_sample, edge, edge_pitch, epoch, ctx->dc_gave_up);
}

}                               // extern "C"
