// plane_iiwt.cpp -- the C ABI of libschro_hip.so (include/schro_hip.h), plane layer: the inverse wavelet (schro_hip_iiwt_batch:
// the level loop, the register / LDS / Haar forms, the combine form, the transform in two calls; experiments build: fused
// LDS groups and the chain form).  r05: plane.cpp split by entry point.

#include "schro_hip_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

using namespace schro;

extern "C" {

// ---- plane layer ----------------------------------------------------------------

// levels fb .. fb+nl-1 of every plane in one launch of the fused LDS kernel
static int
iiwt_fused_group (SchroHipContext * ctx, const SchroHipIwtPlane * planes, int nplanes, int depth,
    int filter, int bpp, int fb, int nl, const std::vector < size_t > &scratch_off,
    const std::vector < int >&scratch_stride, int uc, int ur)
{
  const size_t jsz = iiwt_fused_job_size ();
  std::vector < char >fj (jsz * nplanes);
  int tile_base = 0;
  for (int p = 0; p < nplanes; p++) {
    const SchroHipIwtPlane & pl = planes[p];
    const int top = fb + nl;    // first level above the group
    const void *ll = pl.src;
    int ll_stride = (pl.src_stride << (top - 1)) * 2;
    if (top < depth) {
      ll = (const char *) ctx->scratch_ref () + scratch_off[(size_t) p * depth + top];
      ll_stride = scratch_stride[(size_t) p * depth + top];
    }
    void *dst = pl.dst;
    int dst_stride = pl.dst_stride;
    if (fb > 0) {
      dst = (char *) ctx->scratch_ref () + scratch_off[(size_t) p * depth + fb];
      dst_stride = scratch_stride[(size_t) p * depth + fb];
    }
    const int w = pl.width >> fb, h = pl.height >> fb;
    int tiles_x = div_up (w / 2, uc);
    // the level-fb view of the coefficient frame: {w, h, stride << fb}
    iiwt_fused_job_fill (fj.data () + jsz * p, pl.src, pl.src_stride << fb, bpp, nl, ll, ll_stride,
        dst, dst_stride, w, h, tiles_x, tile_base);
    tile_base += tiles_x * div_up (h / 2, ur);
  }
  void *d_jobs;
  int r = push_args (ctx, fj.data (), fj.size (), &d_jobs);
  if (r)
    return r;
  ProfileScope ps (ctx, fb == 0 ? SCHRO_HIP_KERNEL_IIWT_FINEST : SCHRO_HIP_KERNEL_IIWT_COARSE);
  return launch_iiwt_fused (ctx->stream, d_jobs, nplanes, tile_base, filter, bpp, nl);
}

}                               // extern "C"

// ---- r04: the chain form of the register wavelet: one launch for all levels (iiwt_reg.hip) -----------------
// Builds the jobs (plane, level) with their producer / consumer links and counters, and the ONE order in which
// the launch's workgroups take the tiles of all levels:
//   * key of a tile = the first picture row (in level-0 pixels) of its output + a lag per level; a consumer's
//     key is never below its producers' (the lag of level l - 1 is level l's plus the input rows a consumer tile
//     reaches ahead, (RP - H) 2^l pixels), and inside a band of 64 pixel rows coarser levels come first: a
//     topological order in which the coarse rows run just ahead of the finer rows they feed;
//   * planes are dealt to eight lists (largest first to the shortest list) that are merged four tiles at a
//     time: workgroup b -- XCD b mod 8 by the dispatcher's round robin -- mostly takes tiles of "its" planes,
//     so the rows neighbouring tiles share (lifting halos) meet in one L2, and nothing depends across lists.
// The order depends on the batch's geometry only: cached on the device by a hash of it (four slots per queue).
template < typename JOBFN, typename SMALLFN >
static int
iiwt_chain (SchroHipContext * ctx, int nplanes, int depth, int filter, JOBFN level_job, SMALLFN level_is_small, int *done)
{
  *done = 0;
  std::vector < IwtJob > jobs ((size_t) nplanes * depth);
  std::vector < int >small (depth), RP (depth), UR (depth), Hh (depth);
  for (int l = 0; l < depth; l++) {
    int uc, ur, rmin;
    small[l] = level_is_small (l) ? 1 : 0;
    iiwt_reg_geometry (filter, small[l], &uc, &ur, &rmin);
    UR[l] = ur;
    Hh[l] = rmin - ur;
    RP[l] = rmin + Hh[l];
  }
  int uc0, ur0, rmin0;
  iiwt_reg_geometry (filter, 0, &uc0, &ur0, &rmin0);   // (the column geometry is the same for both forms)
  int n_ctr = 0;
  long n_tiles = 0;
  uint64_t h = 1469598103934665603ull;
  auto mix = [&h] (uint64_t v) {
    for (int k = 0; k < 8; k++) {
      h ^= (v >> (8 * k)) & 0xff;
      h *= 1099511628211ull;
    }
  };
  mix ((uint64_t) filter | ((uint64_t) depth << 8) | ((uint64_t) nplanes << 16));
  for (int p = 0; p < nplanes; p++)
    for (int l = depth - 1; l >= 0; l--) {
      bool src_al, dst_al;
      IwtJob & j = jobs[(size_t) p * depth + l];
      j = level_job (p, l, &src_al, &dst_al);
      const int nc = j.w / 2, nr = j.h / 2;
      if (!(src_al && dst_al && nc % 4 == 0 && nr >= RP[l] - Hh[l]))
        return 0;               // some level of some plane needs another kernel: a launch per level
      j.tiles_x = div_up (nc, uc0);
      const int tiles_y = div_up (nr, UR[l]);
      if ((long) j.tiles_x * tiles_y > 0xffff)
        return 0;
      j.tile_base = tiles_y;    // (no tile bases in this form: the field carries the job's tile rows to the order builder)
      j.small = small[l];
      if (l > 0) {
        j.ctr = n_ctr;
        n_ctr += tiles_y;
      }
      mix ((uint64_t) j.tiles_x);
      if (l < depth - 1) {
        const IwtJob & prod = jobs[(size_t) p * depth + l + 1];
        j.dep_rows2 = 2 * UR[l + 1];
        j.dep_tiles_y = prod.tile_base;
        j.dep_tiles_x = prod.tiles_x;
        j.dep_ctr = prod.ctr;
      }
      n_tiles += (long) j.tiles_x * tiles_y;
      mix ((uint64_t) j.w | ((uint64_t) j.h << 20) | ((uint64_t) small[l] << 40));
    }
  if (jobs.size () > 0xffff || n_tiles > (1L << 24))
    return 0;
  (void) hipSetDevice (ctx->device);

  // ---- the order, cached by geometry ----
  constexpr int per_queue = SchroHipContext::kChainSlots / SchroHipContext::kQueues;
  const int k0 = ctx->cur * per_queue;
  SchroHipContext::ChainSlot * slot = nullptr, *lru = &ctx->chain_slots[k0];
  for (int k = k0; k < k0 + per_queue; k++) {
    SchroHipContext::ChainSlot & o = ctx->chain_slots[k];
    if (o.d && o.hash == h && o.count == (size_t) n_tiles)
      slot = &o;
    if (o.last_use < lru->last_use)
      lru = &o;
  }
  if (!slot) {
    std::vector < long >lag (depth, 0);
    for (int l = depth - 1; l >= 1; l--)
      lag[l - 1] = lag[l] + (long) (RP[l - 1] - Hh[l - 1]) * (1L << l);
    struct Key {
      long band;
      int level;
      uint32_t entry;
    };
    // planes to lists: largest first to the shortest list
    constexpr int kLists = 8;
    std::vector < int >by_size (nplanes);
    std::vector < long >plane_tiles (nplanes, 0);
    for (int p = 0; p < nplanes; p++) {
      by_size[p] = p;
      for (int l = 0; l < depth; l++)
        plane_tiles[p] += (long) jobs[(size_t) p * depth + l].tiles_x * jobs[(size_t) p * depth + l].tile_base;
    }
    std::stable_sort (by_size.begin (), by_size.end (),[&](int a, int b) { return plane_tiles[a] > plane_tiles[b]; });
    std::vector < Key > lists[kLists];
    long load[kLists] = { 0 };
    for (int p : by_size) {
      int best = 0;
      for (int k = 1; k < kLists; k++)
        if (load[k] < load[best])
          best = k;
      load[best] += plane_tiles[p];
      for (int l = 0; l < depth; l++) {
        const size_t ji = (size_t) p * depth + l;
        const IwtJob & j = jobs[ji];
        const int nr = j.h / 2, tiles_y = j.tile_base;
        for (int ty = 0; ty < tiles_y; ty++) {
          int r0 = ty * UR[l] - Hh[l];
          if (r0 + Hh[l] + UR[l] > nr)
            r0 = nr - UR[l] - Hh[l];
          const long key = ((long) (2 * (r0 + Hh[l])) << l) + lag[l];
          for (int tx = 0; tx < j.tiles_x; tx++)
            lists[best].push_back (Key { key / 64, l, (uint32_t) (ji << 16) | (uint32_t) (ty * j.tiles_x + tx) });
        }
      }
    }
    // (experiments: SCHRO_HIP_IIWT_CHAIN_ORDER=band interleaves the levels band by band -- consumers right behind
    // their producers: 8 x 2160p 0.218 ms against 0.104 for a launch per level, the waves in flight are mostly
    // consumers polling; level-major hands out a level's tiles when the level above is long under way)
    static const bool by_band = SCHRO_ENV ("SCHRO_HIP_IIWT_CHAIN_ORDER") && !strcmp (SCHRO_ENV ("SCHRO_HIP_IIWT_CHAIN_ORDER"), "band");
    for (auto & L : lists)
      std::stable_sort (L.begin (), L.end (),[](const Key & a, const Key & b) {
            if (by_band)
              return a.band != b.band ? a.band < b.band : a.level > b.level;
            return a.level != b.level ? a.level > b.level : a.band < b.band;
          });
    std::vector < uint32_t > order;
    order.reserve ((size_t) n_tiles);
    size_t at[kLists] = { 0 };
    while (order.size () < (size_t) n_tiles)
      for (int k = 0; k < kLists; k++)
        for (int n = 0; n < 4 && at[k] < lists[k].size (); n++)
          order.push_back (lists[k][at[k]++].entry);
    slot = lru;
    // (a slot's old table may still be read by launches in flight on this queue)
    SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
    if (slot->cap < order.size ()) {
      if (slot->d)
        SCHRO_HIP_CHECK (hipFree (slot->d));
      slot->d = nullptr;
      slot->cap = 0;
      const size_t cap = order.size () + order.size () / 4;
      SCHRO_HIP_CHECK (hipMalloc ((void **) &slot->d, cap * sizeof (uint32_t)));
      slot->cap = cap;
    }
    SCHRO_HIP_CHECK (hipMemcpy (slot->d, order.data (), order.size () * sizeof (uint32_t), hipMemcpyHostToDevice));
    slot->hash = h;
    slot->count = order.size ();
  }
  slot->last_use = ++ctx->arg_clock;

  // ---- the give-up word; the counters of this queue: they count on from launch to launch of one geometry ----
  if (!ctx->dc_gave_up) {
    SCHRO_HIP_CHECK (hipHostMalloc ((void **) &ctx->dc_gave_up, 64, hipHostMallocDefault));
    memset (ctx->dc_gave_up, 0, 64);
  }
  const bool gave_up_before = ((volatile uint32_t *) ctx->dc_gave_up)[1] != 0;
  {
    const int r = dc_gave_up (ctx);
    if (r) {
      ctx->chain_ctrl_hash[ctx->cur] = 0;       // (the counters of the launch that gave up are short)
      return r;
    }
  }
  const size_t ctrl_words = (size_t) std::max (n_ctr, 1);
  const int q = ctx->cur;
  if (ctx->chain_ctrl_words[q] < ctrl_words) {
    if (ctx->chain_ctrl[q]) {
      SCHRO_HIP_CHECK (hipStreamSynchronize (ctx->stream));
      SCHRO_HIP_CHECK (hipFree (ctx->chain_ctrl[q]));
      ctx->chain_ctrl[q] = nullptr;
      ctx->chain_ctrl_words[q] = 0;
    }
    const size_t cap = ctrl_words + ctrl_words / 2 + 64;
    SCHRO_HIP_CHECK (hipMalloc ((void **) &ctx->chain_ctrl[q], cap * sizeof (uint32_t)));
    ctx->chain_ctrl_words[q] = cap;
    ctx->chain_ctrl_hash[q] = 0;
  }
  int max_tx = 1;
  for (const auto & j : jobs)
    max_tx = std::max (max_tx, j.tiles_x);
  if (ctx->chain_ctrl_hash[q] != h || (uint64_t) (ctx->chain_runs[q] + 2) * (uint64_t) max_tx > 0x7fffffffull || gave_up_before) {
    SCHRO_HIP_CHECK (hipMemsetAsync (ctx->chain_ctrl[q], 0, ctx->chain_ctrl_words[q] * sizeof (uint32_t), ctx->stream));
    ctx->chain_ctrl_hash[q] = h;
    ctx->chain_runs[q] = 0;
  }
  const uint32_t run = ++ctx->chain_runs[q];
  if (++ctx->dc_epoch == 0)
    ctx->dc_epoch = 1;          // (0 = "nothing gave up"; the DC kernel's tags are its own business: dc_edge_for)
  for (auto & j : jobs)
    j.tile_base = 0;
  void *d_jobs;
  int r = push_args (ctx, jobs.data (), sizeof (IwtJob) * jobs.size (), &d_jobs);
  if (r)
    return r;
  {
    ProfileScope ps (ctx, SCHRO_HIP_KERNEL_IIWT_FINEST);
    r = launch_iiwt_chain (ctx->stream, (const IwtJob *) d_jobs, slot->d, (int) n_tiles, ctx->chain_ctrl[q], run,
        ctx->dc_gave_up + 1, ctx->dc_epoch, filter);
  }
  *done = 1;
  return r;
}

extern "C" {

// r04: the planes of a batch whose combine could not be the register kernel's last step: residual plane in the
// scratch + prediction (or + 128) -> picture, by the convert kernel
static int
iiwt_combine_temps (SchroHipContext * ctx, const SchroHipIwtPlane * planes, int nplanes, int depth, int bpp,
    const std::vector < size_t > &scratch_off, const std::vector < int >&scratch_stride)
{
  int tw, th;
  convert_tile_geometry (&tw, &th);
  std::vector < ConvertJob > cj;
  int tile_base = 0;
  for (int p = 0; p < nplanes; p++) {
    const SchroHipIwtPlane & pl = planes[p];
    if (!pl.combine || !scratch_stride[(size_t) p * depth])
      continue;
    ConvertJob j;
    memset (&j, 0, sizeof (j));
    j.src = (const char *) ctx->scratch_ref () + scratch_off[(size_t) p * depth];
    j.src_stride = scratch_stride[(size_t) p * depth];
    j.dst = (uint8_t *) pl.dst;
    j.dst_stride = pl.dst_stride;
    j.w = pl.out_width;
    j.h = pl.out_height;
    j.pred = pl.combine == 1 ? pl.pred : nullptr;
    j.pred_stride = pl.pred_stride;
    j.tiles_x = div_up (j.w, tw);
    j.tile_base = tile_base;
    tile_base += j.tiles_x * div_up (j.h, th);
    cj.push_back (j);
  }
  if (cj.empty ())
    return 0;
  void *d_jobs;
  int r = push_args (ctx, cj.data (), sizeof (ConvertJob) * cj.size (), &d_jobs);
  if (r)
    return r;
  ProfileScope ps (ctx, SCHRO_HIP_KERNEL_CONVERT);
  return launch_convert (ctx->stream, (const ConvertJob *) d_jobs, (int) cj.size (), tile_base, bpp);
}

int
schro_hip_iiwt_batch (SchroHipContext * ctx, const SchroHipIwtPlane * planes, int nplanes,
    int depth, int filter, int bpp)
{
  SCHRO_HIP_REQUIRE (ctx && planes && nplanes > 0, "iiwt_batch: bad arguments");
  SCHRO_HIP_REQUIRE (nplanes <= kMaxJobs, "iiwt_batch: at most %d planes per call", kMaxJobs);
  SCHRO_HIP_REQUIRE (depth >= 1 && depth <= 6, "iiwt_batch: transform depth %d out of range", depth);
  SCHRO_HIP_REQUIRE (filter >= 0 && filter <= 6, "iiwt_batch: wavelet filter index %d out of range",
      filter);
  SCHRO_HIP_REQUIRE (bpp == 2 || bpp == 4, "iiwt_batch: bpp must be 2 or 4");
  (void) hipSetDevice (ctx->device);

  // per level: planes that allow it run the register form (iiwt_reg.hip), the rest the
  // LDS tile kernel; SCHRO_HIP_IIWT_REG=0 keeps everything on the LDS kernel
  const bool use_reg = iiwt_reg_supported (filter, bpp)
      && !(SCHRO_ENV ("SCHRO_HIP_IIWT_REG") && atoi (SCHRO_ENV ("SCHRO_HIP_IIWT_REG")) == 0);
  int ruc = 0, rur = 0, rmin = 0, rmin_any = 0;
  if (use_reg) {
    int suc, sur, srmin;
    iiwt_reg_geometry (filter, 0, &ruc, &rur, &rmin);
    iiwt_reg_geometry (filter, 1, &suc, &sur, &srmin);
    rmin_any = std::max (rmin, srmin);  // (whichever tile form the level loop picks for level 0)
  }

  // scratch for the intermediate LL images: levels depth-1 .. 1 of every plane
  std::vector < size_t > scratch_off ((size_t) nplanes * depth, 0);
  std::vector < int >scratch_stride ((size_t) nplanes * depth, 0);
  size_t total = 0;
  bool any_ll = false;
  for (int p = 0; p < nplanes; p++) {
    const SchroHipIwtPlane & pl = planes[p];
    SCHRO_HIP_REQUIRE (pl.src && pl.dst, "iiwt_batch: plane %d has a NULL pointer", p);
    SCHRO_HIP_REQUIRE (pl.width > 0 && pl.height > 0 && pl.width % (1 << depth) == 0
        && pl.height % (1 << depth) == 0,
        "iiwt_batch: plane %d size %dx%d is not a multiple of 2^depth", p, pl.width, pl.height);
    // r04, the combine form: dst is the u8 PICTURE (out_width x out_height inside the transform's size)
    SCHRO_HIP_REQUIRE (pl.combine >= 0 && pl.combine <= 2 && (pl.combine == 0 || (pl.out_width > 0 && pl.out_height > 0
                && pl.out_width <= pl.width && pl.out_height <= pl.height)) && (pl.combine != 1 || (pl.pred && pl.pred_stride >= pl.out_width)),
        "iiwt_batch: plane %d: combine %d needs out_width x out_height inside the transform (and a prediction plane for 1)", p, pl.combine);
    SCHRO_HIP_REQUIRE (pl.src_stride >= pl.width * bpp && pl.dst_stride >= (pl.combine ? pl.out_width : pl.width * bpp),
        "iiwt_batch: plane %d stride too small", p);
    // r04: the coarsest level's LL band from a plane of the caller's (the transform split in two calls)
    SCHRO_HIP_REQUIRE (!pl.ll || (pl.ll_stride >= (pl.width >> depth) * bpp && ((uintptr_t) pl.ll | (uintptr_t) pl.ll_stride) % bpp == 0),
        "iiwt_batch: plane %d: the LL plane's stride %d does not hold %d samples", p, pl.ll_stride, pl.width >> depth);
    any_ll |= pl.ll != nullptr;
    {
      const char *s0 = (const char *) pl.src, *s1 = s0 + (size_t) pl.src_stride * pl.height;
      const char *d0 = (const char *) pl.dst, *d1 = d0 + (size_t) pl.dst_stride * (pl.combine ? pl.out_height : pl.height);
      SCHRO_HIP_REQUIRE (s1 <= d0 || d1 <= s0, "iiwt_batch: plane %d src and dst overlap", p);
    }
    // a combine plane whose finest level cannot take the register kernel's combine form (s32, the fidelity
    // filter, unaligned planes) goes through a residual plane in the scratch and the convert kernel
    if (pl.combine) {
      // (exactly what the level loop below asks of a level-0 register tile -- ADVICE r04: a plane this test let
      // through and the loop then refused returned EINVAL instead of taking the scratch route)
      const bool direct = bpp == 2 && use_reg && (pl.width / 2) % 4 == 0 && pl.height / 2 >= rmin_any
          && ((((uintptr_t) pl.src | (uintptr_t) pl.src_stride | (uintptr_t) pl.dst | (uintptr_t) pl.dst_stride) & 7) == 0)
          // (a depth-1 call reads its LL band from the caller's plane)
          && (depth > 1 || !pl.ll || (((uintptr_t) pl.ll | (uintptr_t) pl.ll_stride) & 7) == 0)
          // (the prediction's rows: 8-byte aligned and readable up to a multiple of 8 columns)
          && (pl.combine != 1 || ((((uintptr_t) pl.pred | (uintptr_t) pl.pred_stride) & 7) == 0 && pl.pred_stride >= ((pl.out_width + 7) & ~7)));
      if (!direct) {
        int stride = (int) round_up ((size_t) pl.width * bpp, 128);
        scratch_off[(size_t) p * depth] = total;
        scratch_stride[(size_t) p * depth] = stride;
        total += round_up ((size_t) stride * pl.height, 256);
      }
    }
    for (int l = 1; l < depth; l++) {
      int w = pl.width >> l, h = pl.height >> l;
      // (whole 128-byte lines per row: in the chain form of the register kernels a consumer tile must never bring a
      // line into its XCD's L2 that also holds samples of a row its producer has not written yet)
      int stride = (int) round_up ((size_t) w * bpp, 128);
      scratch_off[(size_t) p * depth + l] = total;
      scratch_stride[(size_t) p * depth + l] = stride;
      total += round_up ((size_t) stride * h, 256);
    }
  }
  if (total) {
    int r = ensure_scratch (ctx, total);
    if (r)
      return r;
  }

  int uc, ur;
  iiwt_tile_geometry (filter, bpp, &uc, &ur);


  // Fused group (opt-in): SCHRO_HIP_IIWT_FUSE=n runs levels b .. b+n-1 as ONE launch of
  // the fused LDS kernel, b = SCHRO_HIP_IIWT_FUSE_BASE (default 1 where level 0 has the
  // register kernel, else 0).  It saves launches and the intermediate LL round trips, but
  // measured on 8 x 2160p it loses to a launch per level both for the finest levels
  // (0.181 vs 0.166 ms, LDS kernels) and for levels 2+1 (0.054 vs 0.042 ms against the
  // register kernel), so nothing is fused by default.
  int fb = 0, nl = 0;
  {
    const char *env = SCHRO_ENV ("SCHRO_HIP_IIWT_FUSE"), *envb = SCHRO_ENV ("SCHRO_HIP_IIWT_FUSE_BASE");
    fb = envb ? atoi (envb) : (use_reg ? 1 : 0);
    fb = std::max (0, std::min (fb, depth - 1));
    int want = env ? atoi (env) : 0;
    for (int p = 0; p < nplanes; p++)
      if (planes[p].combine || planes[p].ll)
        want = 0;               // (the combine form and split transforms belong to the per-level kernels)
    nl = std::min (std::min (want, depth - fb), iiwt_fused_max_levels (filter, bpp));
    const int vl = 8 / bpp;
    for (int p = 0; p < nplanes && nl >= 2; p++) {
      const SchroHipIwtPlane & pl = planes[p];
      if ((((uintptr_t) pl.src | (uintptr_t) pl.src_stride) & 7) != 0)
        nl = 0;
      while (nl >= 2 && (((pl.width >> (fb + nl)) % vl) != 0 || (pl.width >> (fb + nl)) < vl))
        nl--;
    }
    if (nl < 2)
      nl = 0;
  }

  // s32 Haar levels (the low-delay 10-bit configurations): the element-wise form of iiwt_haar.hip
  const bool use_haar = iiwt_haar_supported (filter, bpp)
      && !(SCHRO_ENV ("SCHRO_HIP_IIWT_HAAR") && atoi (SCHRO_ENV ("SCHRO_HIP_IIWT_HAAR")) == 0);
  int hcols = 1, hrows = 1;
  if (use_haar)
    iiwt_haar_geometry (&hcols, &hrows);

  // r03: a depth-3 s32 Haar transform (the low-delay 10-bit configurations) is ONE pass over the
  // coefficient frame when every plane allows it (iiwt_haar.hip, iiwt_haar3_s32_kernel);
  // SCHRO_HIP_IIWT_HAAR3=0 keeps a launch per level
  if (use_haar && depth == 3 && !nl && !any_ll && !(SCHRO_ENV ("SCHRO_HIP_IIWT_HAAR3") && atoi (SCHRO_ENV ("SCHRO_HIP_IIWT_HAAR3")) == 0)) {
    bool all_ok = true;
    for (int p = 0; p < nplanes && all_ok; p++)
      all_ok = iiwt_haar3_job_ok (planes[p].src, planes[p].src_stride, planes[p].dst, planes[p].dst_stride, planes[p].width,
          planes[p].height);
    if (all_ok) {
      int bxs, bys;
      iiwt_haar3_geometry (&bxs, &bys);
      std::vector < IwtJob > j3 (nplanes);
      int tile_base = 0;
      for (int p = 0; p < nplanes; p++) {
        IwtJob & j = j3[p];
        memset (&j, 0, sizeof (j));
        j.sb[0] = planes[p].src;
        j.sb_stride[0] = planes[p].src_stride;
        j.dst = planes[p].dst;
        j.dst_stride = planes[p].dst_stride;
        if (planes[p].combine) {        // (s32: always through a residual plane in the scratch)
          j.dst = (char *) ctx->scratch_ref () + scratch_off[(size_t) p * depth];
          j.dst_stride = scratch_stride[(size_t) p * depth];
        }
        j.w = planes[p].width;
        j.h = planes[p].height;
        j.tiles_x = div_up (j.w / 8, bxs);
        j.tile_base = tile_base;
        tile_base += j.tiles_x * div_up (j.h / 8, bys);
      }
      void *d_j3;
      int r = push_args (ctx, j3.data (), sizeof (IwtJob) * j3.size (), &d_j3);
      if (r)
        return r;
      {
        ProfileScope ps (ctx, SCHRO_HIP_KERNEL_IIWT_FINEST);
        r = launch_iiwt_haar3 (ctx->stream, (const IwtJob *) d_j3, nplanes, tile_base, filter);
      }
      return r ? r : iiwt_combine_temps (ctx, planes, nplanes, depth, bpp, scratch_off, scratch_stride);
    }
  }

  // the job of (plane, level): the level view of the coefficient frame {w, h, stride << level}
  // (schrodecoder.c:1834-1845), sub-band positions schroparams.c:319-352, LL from / output to the scratch
  auto level_job = [&](int p, int level, bool * src_al_out, bool * dst_al_out) {
    const SchroHipIwtPlane & pl = planes[p];
    IwtJob j;
    memset (&j, 0, sizeof (j));
    int w = pl.width >> level, h = pl.height >> level;
    const char *base = (const char *) pl.src;
    int vstride = pl.src_stride << level;
    const char *ll = base;
    int ll_stride = vstride * 2;
    if (level < depth - 1) {
      ll = (const char *) ctx->scratch_ref () + scratch_off[(size_t) p * depth + level + 1];
      ll_stride = scratch_stride[(size_t) p * depth + level + 1];
    } else if (pl.ll) {         // (r04: the coarser levels ran in a call of their own)
      ll = (const char *) pl.ll;
      ll_stride = pl.ll_stride;
    }
    j.sb[0] = ll;
    j.sb_stride[0] = ll_stride;
    j.sb[1] = base + (size_t) (w / 2) * bpp;
    j.sb_stride[1] = vstride * 2;
    j.sb[2] = base + vstride;
    j.sb_stride[2] = vstride * 2;
    j.sb[3] = base + vstride + (size_t) (w / 2) * bpp;
    j.sb_stride[3] = vstride * 2;
    if (level == 0 && pl.combine && scratch_stride[(size_t) p * depth]) {
      // (combine through a residual plane in the scratch: see above)
      j.dst = (char *) ctx->scratch_ref () + scratch_off[(size_t) p * depth];
      j.dst_stride = scratch_stride[(size_t) p * depth];
    } else if (level == 0) {
      j.dst = pl.dst;
      j.dst_stride = pl.dst_stride;
      if (pl.combine) {
        j.pred = pl.combine == 1 ? pl.pred : nullptr;
        j.pred_stride = pl.pred_stride;
        j.out_w = pl.out_width;
        j.out_h = pl.out_height;
        j.pad2 = 1;             // (combine form)
      }
    } else {
      j.dst = (char *) ctx->scratch_ref () + scratch_off[(size_t) p * depth + level];
      j.dst_stride = scratch_stride[(size_t) p * depth + level];
    }
    j.w = w;
    j.h = h;
    int nc = w / 2;
    int vl = 8 / bpp;
    bool src_al = (nc % vl) == 0 && nc >= vl;
    for (int s = 0; s < 4; s++)
      src_al = src_al && (((uintptr_t) j.sb[s] | (uintptr_t) j.sb_stride[s]) & 7) == 0;
    bool dst_al = (((uintptr_t) j.dst | (uintptr_t) j.dst_stride) & (j.pad2 ? 7 : 15)) == 0;
    j.flags = (src_al ? 1 : 0) | (dst_al ? 2 : 0);
    j.ctr = -1;
    *src_al_out = src_al;
    *dst_al_out = dst_al;
    return j;
  };
  // which tile form a level takes in the register kernels: a level of fewer tiles than the chip has SIMDs
  // twice over is latency, not bandwidth -- the small form (4 useful row pairs per wave)
  auto level_is_small = [&](int level) {
    long tiles = 0;
    for (int p = 0; p < nplanes; p++)
      tiles += (long) div_up ((planes[p].width >> level) / 2, ruc) * div_up ((planes[p].height >> level) / 2, rur);
    const char *env = SCHRO_ENV ("SCHRO_HIP_IIWT_SMALL");
    const char *envb = SCHRO_ENV ("SCHRO_HIP_IIWT_SMALL_BELOW");
    // (r03, SCHRO_HIP_IIWT_SMALL_BELOW: with 4096 the 3264-tile level -- 8 x 1080p's finest, 8 x 2160p's
    // middle one -- takes the small form: alone 0.0381 -> 0.0351 ms, but 0.0263 -> 0.0322 with two batches in
    // flight, and 0.002 ms of a 2160p step: left at 2048)
    return env ? atoi (env) != 0 : tiles < (envb ? atol (envb) : 2048);
  };

  // r04: every level in ONE launch where all of them can take the register form (iiwt_reg.hip, chain form).
  // Built as VERDICT r03 asked and measured: 8 x 2160p 0.135 ms against 0.103 for a launch per level (8 x 1080p
  // 0.058 against 0.038) -- a tile's extra round trip to its producers' counters and the written-through LL
  // stores cost more than the launch gaps they remove -- so it is opt-in: SCHRO_HIP_IIWT_CHAIN=1
  bool any_combine = false;
  for (int p = 0; p < nplanes; p++)
    any_combine |= planes[p].combine != 0;
  if (use_reg && depth >= 2 && !nl && !any_combine && !any_ll && SCHRO_ENV ("SCHRO_HIP_IIWT_CHAIN") && atoi (SCHRO_ENV ("SCHRO_HIP_IIWT_CHAIN")) != 0) {
    int done = 0;
    const int r = iiwt_chain (ctx, nplanes, depth, filter, level_job, level_is_small, &done);
    if (r || done)
      return r;
  }

  std::vector < IwtJob > jobs, rjobs, hjobs, cjobs;
  for (int level = depth - 1; level >= 0; level--) {
    if (nl && level >= fb && level < fb + nl) {
      if (level == fb + nl - 1) {
        int r = iiwt_fused_group (ctx, planes, nplanes, depth, filter, bpp, fb, nl, scratch_off, scratch_stride, uc, ur);
        if (r)
          return r;
      }
      continue;
    }
    int tile_base = 0, rtile_base = 0, htile_base = 0, ctile_base = 0;
    jobs.clear ();
    rjobs.clear ();
    hjobs.clear ();
    cjobs.clear ();
    int lruc = ruc, lrur = rur, lrmin = rmin, small = 0;
    if (use_reg) {
      small = level_is_small (level);
      if (small)
        iiwt_reg_geometry (filter, 1, &lruc, &lrur, &lrmin);
    }
    for (int p = 0; p < nplanes; p++) {
      bool src_al, dst_al;
      IwtJob j = level_job (p, level, &src_al, &dst_al);
      const int nc = j.w / 2, nr = j.h / 2;
      if (use_haar && iiwt_haar_job_ok (j)) {
        j.tiles_x = div_up (nc, hcols);
        j.tile_base = htile_base;
        htile_base += j.tiles_x * div_up (nr, hrows);
        hjobs.push_back (j);
      } else if (use_reg && src_al && dst_al && nc % 4 == 0 && nr >= lrmin) {
        j.tiles_x = div_up (nc, lruc);
        if (j.pad2) {           // the combine form: its own launch (another instantiation of the kernel)
          j.tile_base = ctile_base;
          ctile_base += j.tiles_x * div_up (nr, lrur);
          cjobs.push_back (j);
        } else {
          j.tile_base = rtile_base;
          rtile_base += j.tiles_x * div_up (nr, lrur);
          rjobs.push_back (j);
        }
      } else if (j.pad2) {
        return set_error (SCHRO_HIP_EINVAL, "iiwt_batch: plane %d: the combine form was promised a register tile it cannot have", p);
      } else {
        j.tiles_x = div_up (nc, uc);
        j.tile_base = tile_base;
        tile_base += j.tiles_x * div_up (nr, ur);
        jobs.push_back (j);
      }
    }
    void *d_rjobs = nullptr, *d_jobs = nullptr, *d_hjobs = nullptr, *d_cjobs = nullptr;
    int r = 0;
    if (!rjobs.empty ())
      r = push_args (ctx, rjobs.data (), sizeof (IwtJob) * rjobs.size (), &d_rjobs);
    if (!r && !cjobs.empty ())
      r = push_args (ctx, cjobs.data (), sizeof (IwtJob) * cjobs.size (), &d_cjobs);
    if (!r && !hjobs.empty ())
      r = push_args (ctx, hjobs.data (), sizeof (IwtJob) * hjobs.size (), &d_hjobs);
    if (!r && !jobs.empty ())
      r = push_args (ctx, jobs.data (), sizeof (IwtJob) * jobs.size (), &d_jobs);
    if (r)
      return r;
    ProfileScope ps (ctx, level == 0 ? SCHRO_HIP_KERNEL_IIWT_FINEST : SCHRO_HIP_KERNEL_IIWT_COARSE);
    if (d_rjobs)
      r = launch_iiwt_reg (ctx->stream, (const IwtJob *) d_rjobs, (int) rjobs.size (), rtile_base, filter, small, 0);
    if (!r && d_cjobs)
      r = launch_iiwt_reg (ctx->stream, (const IwtJob *) d_cjobs, (int) cjobs.size (), ctile_base, filter, small, 1);
    if (!r && d_hjobs)
      r = launch_iiwt_haar (ctx->stream, (const IwtJob *) d_hjobs, (int) hjobs.size (), htile_base, filter);
    if (!r && d_jobs)
      r = launch_iiwt_level (ctx->stream, (const IwtJob *) d_jobs, (int) jobs.size (), tile_base, filter, bpp);
    if (r)
      return r;
  }
  return iiwt_combine_temps (ctx, planes, nplanes, depth, bpp, scratch_off, scratch_stride);
}

}                               // extern "C"
