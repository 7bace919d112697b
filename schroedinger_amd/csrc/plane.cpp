// plane.cpp -- the C ABI of libschro_hip.so (include/schro_hip.h), second part: the batched plane-level launches
// (inverse wavelet, convert, pack, low-delay slices, DC prediction, dequantisation and its plans, upsample, OBMC)
// -- what a host that owns device memory binds, and what the frame layer (frame.cpp) is built on.

#include "schro_hip_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

using namespace schro;

static_assert (sizeof (ObmcJob) * kMaxJobs <= SchroHipContext::kArgSlotBytes, "a launch group of kMaxJobs OBMC jobs fits a table slot");

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
  return launch_upsample (ctx->stream, (const UpsampleJob *) d_jobs, nplanes, tile_base);
}

// The order in which an OBMC launch walks its tiles.  Workgroups go to the 8 XCDs round
// robin and xcd_tile_id () gives every XCD one contiguous run of positions; with the plain
// plane-by-plane list such a run is about one picture of a batch, so every XCD pulls BOTH
// reference images of every picture through its own 4 MiB L2 -- for the 8 pictures between
// two anchors, which share their references, 8 times the same 100 MB.  Here position v of the
// order holds tile (job << 16 | tile): sorted by the tile's vertical position in its plane,
// then by reference, so an XCD's run is a horizontal stripe of ALL the pictures and the tiles
// that read the same reference rows follow each other.
//
// The table depends on the launch's tile geometry and on WHICH jobs share a reference, not
// on where the references live: "reference" is the index of the first job of the launch with
// the same first reference, so a decoder whose reference frames move through a pool from
// GOP to GOP finds its table again.  A new table goes to the device with an asynchronous
// copy from the slot's pinned mirror on the context's stream -- behind the kernels that
// still read the slot's old table, ahead of the launch that wants the new one; the stream
// is never drained here.
// scratch runs only: SCHRO_HIP_OBMC_STAMPS=1 gives the staged kernel a buffer for per-phase
// cycle stamps; schro_hip_obmc_stamps_dump () prints their medians
// SCHRO_HIP_OBMC_MERGE=0: every plane its own job in the row kernel (A/B runs); 2: pairs always
static int
obmc_row_merge_mode ()
{
  // 0: never, 1: where it pays (default), 2: always (the tests run the pair kernels on small pictures)
  static const int mode = SCHRO_ENV ("SCHRO_HIP_OBMC_MERGE") ? atoi (SCHRO_ENV ("SCHRO_HIP_OBMC_MERGE")) : 1;
  return mode;
}

static unsigned long long *g_stamps;
static unsigned long long *
obmc_stamp_buffer ()
{
  static const bool on = SCHRO_ENV ("SCHRO_HIP_OBMC_STAMPS") != nullptr;
  if (on && !g_stamps) {
    if (hipMalloc ((void **) &g_stamps, 16384 * 16 * 8) != hipSuccess)
      g_stamps = nullptr;
    else
      (void) hipMemset (g_stamps, 0, 16384 * 16 * 8);
  }
  return g_stamps;
}

extern "C" void
schro_hip_obmc_stamps_dump (void)
{
  if (!g_stamps)
    return;
  (void) hipDeviceSynchronize ();
  std::vector < unsigned long long >h (16384 * 16);
  (void) hipMemcpy (h.data (), g_stamps, h.size () * 8, hipMemcpyDeviceToHost);
  for (int n = 1; n <= 11; n++) {
    std::vector < unsigned long long >v;
    for (int b = 0; b < 16384; b++)
      if (h[b * 16 + 9])
        v.push_back (h[b * 16 + n]);
    if (v.empty ())
      continue;
    std::sort (v.begin (), v.end ());
    unsigned long long sum = 0;
    for (auto x : v)
      sum += x;
    fprintf (stderr, "stamp %d: median %llu  p10 %llu  p90 %llu  p99 %llu  max %llu  mean %llu  (n=%zu)\n", n, v[v.size () / 2],
        v[v.size () / 10], v[v.size () * 9 / 10], v[v.size () * 99 / 100], v.back (), sum / v.size (), v.size ());
  }
  // occupancy: workgroup lifetimes against the span of the workgroups that ran on the same CU
  // (HW_ID: cu 8-11, sh 12, se 13-15; XCC_ID 0-3)
  std::vector < std::pair < int, int > >by_cu;
  for (int b = 0; b < 16384; b++)
    if (h[b * 16 + 9])
      by_cu.push_back ({ (int) (((h[b * 16 + 14] >> 8) & 0xff) | ((h[b * 16 + 15] & 0xf) << 8)), b });
  std::sort (by_cu.begin (), by_cu.end ());
  double life_all = 0, cap_all = 0;
  unsigned long long span_max = 0, span_min = ~0ull;
  size_t ncu = 0;
  for (size_t i = 0; i < by_cu.size ();) {
    size_t j = i;
    unsigned long long t0 = ~0ull, t1 = 0, life = 0;
    for (; j < by_cu.size () && by_cu[j].first == by_cu[i].first; j++) {
      const int b = by_cu[j].second;
      t0 = std::min (t0, h[b * 16 + 12]);
      t1 = std::max (t1, h[b * 16 + 13]);
      life += h[b * 16 + 13] - h[b * 16 + 12];
    }
    life_all += (double) life;
    cap_all += (double) (t1 - t0);
    span_max = std::max (span_max, t1 - t0);
    span_min = std::min (span_min, t1 - t0);
    ncu++;
    i = j;
  }
  for (int x = 0; x < 16; x++) {
    unsigned long long t0 = ~0ull, t1 = 0, life = 0;
    size_t nw = 0;
    for (int b = 0; b < 16384; b++)
      if (h[b * 16 + 9] && (int) (h[b * 16 + 15] & 0xf) == x) {
        t0 = std::min (t0, h[b * 16 + 12]);
        t1 = std::max (t1, h[b * 16 + 13]);
        life += h[b * 16 + 13] - h[b * 16 + 12];
        nw++;
      }
    if (nw)
      fprintf (stderr, "  XCD %d: %zu workgroups, span %llu ticks, mean lifetime %llu\n", x, nw, t1 - t0, life / nw);
  }
  if (ncu)
    fprintf (stderr, "%zu workgroups on %zu CUs: per-CU span %llu .. %llu ticks, resident workgroups per CU %.2f\n",
        by_cu.size (), ncu, span_min, span_max, life_all / cap_all);
}

static int
obmc_tile_order (SchroHipContext * ctx, const std::vector < ObmcJob > &jobs, int variant, int total,
    const uint32_t ** d_order)
{
  *d_order = nullptr;
  static const bool enabled = !SCHRO_ENV ("SCHRO_HIP_OBMC_ORDER") || atoi (SCHRO_ENV ("SCHRO_HIP_OBMC_ORDER")) != 0;
  if (!enabled || variant < 1 || jobs.empty () || jobs.size () > 0xffff)
    return 0;
  uint64_t h = 1469598103934665603ull;
  auto mix = [&h] (uint64_t v) {
    for (int k = 0; k < 8; k++) {
      h ^= (v >> (8 * k)) & 0xff;
      h *= 1099511628211ull;
    }
  };
  std::vector < int >tiles_y (jobs.size ());
  std::vector < uint32_t > ref_class (jobs.size ());
  mix ((uint64_t) variant);
  for (size_t j = 0; j < jobs.size (); j++) {
    int tx;
    obmc_tiles (variant, jobs[j].w, jobs[j].h, jobs[j].xoff, &tx, &tiles_y[j]);
    if (tx * tiles_y[j] > 0xffff)
      return 0;
    size_t first = 0;
    while (jobs[first].ref[0] != jobs[j].ref[0])
      first++;
    ref_class[j] = (uint32_t) first;
    mix ((uint64_t) tx);
    mix ((uint64_t) tiles_y[j]);
    mix ((uint64_t) first);
  }
  constexpr int per_queue = SchroHipContext::kOrderSlots / SchroHipContext::kQueues;
  const int k0 = ctx->cur * per_queue;
  SchroHipContext::OrderSlot * slot = nullptr, *lru = &ctx->order_slots[k0];
  for (int k = k0; k < k0 + per_queue; k++) {
    SchroHipContext::OrderSlot & o = ctx->order_slots[k];
    if (o.d && o.hash == h && o.count == (size_t) total)
      slot = &o;
    if (o.last_use < lru->last_use)
      lru = &o;
  }
  if (!slot) {
    // r03: the sort unit is a SUPERTILE of 8 x 4 tiles (1024 x 128 pixels), not a row of tiles.  An
    // XCD has up to 32 x 7 = 224 tiles in flight: one supertile of all 8 pictures between two anchors.
    // Their sample windows cover (1024 + 32) x (128 + 44) pixels of each reference, 1.4 MB of the four
    // half-pel planes -- both references fit the XCD's 4 MiB L2 beside the streamed residual.  As rows
    // of tiles (r02) the tiles in flight spanned the picture's width: 2.3 MB per reference with the
    // r03 planes, and the L2 missed 7.1 M lines per step (915 MB) for 205 MB of reference planes.
    struct Key {
      uint32_t row;             // supertile, in raster order over the plane (by relative position: planes of different sizes align)
      uint32_t ref;
      uint32_t entry;
    };
    std::vector < Key > keys;
    keys.reserve ((size_t) total);
    // (late r03: 4 x 4 tiles; with the residual and the picture streamed, 8 x 4 is 1 % behind -- 0.4106 against
    // 0.4056 ms per 8 x 2160p step --, 8 x 8 and 16 x 4 3 - 4 %)
    static const int sup_x = SCHRO_ENV ("SCHRO_HIP_OBMC_SUPER_X") ? std::max (1, atoi (SCHRO_ENV ("SCHRO_HIP_OBMC_SUPER_X"))) : 4;
    static const int sup_y = SCHRO_ENV ("SCHRO_HIP_OBMC_SUPER_Y") ? std::max (1, atoi (SCHRO_ENV ("SCHRO_HIP_OBMC_SUPER_Y"))) : 4;
    for (size_t j = 0; j < jobs.size (); j++) {
      // (a U + V pair reads two planes of each reference: half the width; pairs from pair images -- variant 4,
      // tiles of 64 x 32 chroma pixels -- cover the picture area of a luma tile twice as high: half the height)
      const int sx = variant == 4 ? sup_x : jobs[j].nplanes == 2 ? std::max (1, sup_x / 2) : sup_x;
      const int sy = variant == 4 ? std::max (1, sup_y / 2) : sup_y;
      const int nsx = div_up (jobs[j].tiles_x, sx);
      for (int ty = 0; ty < tiles_y[j]; ty++)
        for (int tx = 0; tx < jobs[j].tiles_x; tx++)
          keys.push_back (Key { (uint32_t) ((ty / sy) * nsx + tx / sx), ref_class[j],
              (uint32_t) (j << 16) | (uint32_t) (ty * jobs[j].tiles_x + tx) });
    }
    if (keys.size () != (size_t) total)
      return set_error (SCHRO_HIP_EINVAL, "obmc tile order: %zu tiles, %d expected", keys.size (), total);
    std::stable_sort (keys.begin (), keys.end (),[](const Key & a, const Key & b) {
          return a.row != b.row ? a.row < b.row : a.ref < b.ref;
        });
    // Every XCD runs a fixed eighth of the workgroups (xcd_tile_id: XCD x runs entries
    // [x q + min (x, r), ...)).  The tiles on the picture's rim take the exact per-sample path
    // for their outer blocks and live two to three times as long as the others (stamps: 57-72 k
    // cycles against a median of 29 k), and the bands at the top and the bottom of the pictures
    // hold most of them: with whole bands per XCD the first and the last XCD ran 25 % longer
    // than the others.  So the rim tiles are dealt out evenly, and each XCD starts with its
    // share of them (longest first), then runs its band of inner tiles.
    {
      auto is_rim = [&](const Key & k) {
        const size_t j = k.entry >> 16;
        const int t = (int) (k.entry & 0xffffu), tx = t % jobs[j].tiles_x, ty = t / jobs[j].tiles_x;
        return tx == 0 || ty == 0 || tx == jobs[j].tiles_x - 1 || ty == tiles_y[j] - 1;
      };
      constexpr size_t kXcd = 8;
      std::vector < Key > rim_sorted, rim, inner;
      for (const Key & k : keys)
        (is_rim (k) ? rim_sorted : inner).push_back (k);
      // (every eighth one to an XCD: top / bottom rows, side columns and corners in equal parts)
      for (size_t x = 0; x < kXcd; x++)
        for (size_t k = x; k < rim_sorted.size (); k += kXcd)
          rim.push_back (rim_sorted[k]);
      const size_t q = keys.size () / kXcd, r = keys.size () % kXcd;
      size_t ri = 0, ii = 0, out = 0;
      for (size_t x = 0; x < kXcd; x++) {
        const size_t n = q + (x < r ? 1 : 0);
        size_t nr = rim.size () / kXcd + (x < rim.size () % kXcd ? 1 : 0);
        nr = std::min (nr, n);
        if (n - nr > inner.size () - ii)        // (more rim than inner tiles: small planes)
          nr = n - (inner.size () - ii);
        for (size_t k = 0; k < nr; k++)
          keys[out++] = rim[ri++];
        for (size_t k = nr; k < n; k++)
          keys[out++] = inner[ii++];
      }
      if (ri != rim.size () || ii != inner.size () || out != keys.size ())
        return set_error (SCHRO_HIP_EINVAL, "obmc tile order: rim / inner split does not add up");
    }
    slot = lru;
    if (slot->copy_pending) {   // the mirror's previous upload: long done unless tables churn
      SCHRO_HIP_CHECK (hipEventSynchronize (slot->copied));
      slot->copy_pending = false;
    }
    if (slot->cap < keys.size ()) {
      // grow-only; hipFree waits for the work that may still read the old table
      if (slot->d)
        SCHRO_HIP_CHECK (hipFree (slot->d));
      if (slot->h)
        SCHRO_HIP_CHECK (hipHostFree (slot->h));
      slot->d = nullptr;
      slot->h = nullptr;
      slot->cap = 0;
      const size_t cap = keys.size () + keys.size () / 4;
      SCHRO_HIP_CHECK (hipMalloc ((void **) &slot->d, cap * sizeof (uint32_t)));
      SCHRO_HIP_CHECK (hipHostMalloc ((void **) &slot->h, cap * sizeof (uint32_t), hipHostMallocDefault));
      slot->cap = cap;
    }
    if (!slot->copied)
      SCHRO_HIP_CHECK (hipEventCreateWithFlags (&slot->copied, hipEventDisableTiming));
    for (size_t k = 0; k < keys.size (); k++)
      slot->h[k] = keys[k].entry;
    SCHRO_HIP_CHECK (hipMemcpyAsync (slot->d, slot->h, keys.size () * sizeof (uint32_t), hipMemcpyHostToDevice,
            ctx->stream));
    SCHRO_HIP_CHECK (hipEventRecord (slot->copied, ctx->stream));
    slot->copy_pending = true;
    slot->hash = h;
    slot->count = keys.size ();
  }
  slot->last_use = ++ctx->arg_clock;
  *d_order = slot->d;
  return 0;
}

int
schro_hip_obmc_batch (SchroHipContext * ctx, const SchroHipObmcPlane * planes, int nplanes)
{
  SCHRO_HIP_REQUIRE (ctx && planes && nplanes > 0 && nplanes <= kMaxJobs,
      "obmc_batch: bad arguments");
  (void) hipSetDevice (ctx->device);
  // kernel variant per plane: default weights (1,1,bits 1) run the LDS-accumulate
  // item kernel, everything else the exact per-pixel kernel
  auto variant_of = [](const SchroHipObmcPlane & pl) {
    return (pl.picture_weight_1 == 1 && pl.picture_weight_2 == 1 && pl.picture_weight_bits == 1) ? 1 : 0;
  };
  // default weights, half- / quarter-pel references and blocks up to 16 wide: the row kernel
  // (obmc_row.hip); SCHRO_HIP_OBMC_KERNEL=item sends them to obmc.hip's item kernel (A/B runs: the
  // second formulation the parity tests compare)
  static const bool use_row = !SCHRO_ENV ("SCHRO_HIP_OBMC_KERNEL") || strcmp (SCHRO_ENV ("SCHRO_HIP_OBMC_KERNEL"), "row") == 0;
  std::vector < ObmcJob > all (nplanes);
  std::vector < int >key (nplanes), row_nd (nplanes);
  uint32_t pred_epoch = 0;      // (r05: this call's number among the context's prediction_only calls, once it has one)
  for (int p = 0; p < nplanes; p++) {
    const SchroHipObmcPlane & pl = planes[p];
    // (residual NULL: nothing to add -- the prediction alone, clamped)
    SCHRO_HIP_REQUIRE (pl.mvs && pl.ref1 && pl.out, "obmc_batch: plane %d has a NULL pointer", p);
    SCHRO_HIP_REQUIRE (pl.mv_precision >= 0 && pl.mv_precision <= 3,
        "obmc_batch: mv_precision %d out of range", pl.mv_precision);
    SCHRO_HIP_REQUIRE (pl.component >= 0 && pl.component <= 2, "obmc_batch: bad component");
    SCHRO_HIP_REQUIRE (!pl.residual || pl.residual_bpp == 2 || pl.residual_bpp == 4, "obmc_batch: residual bpp");
    // r04: the prediction alone, for the wavelet's combine form: it must fit the u8 plane it is written to
    SCHRO_HIP_REQUIRE (!pl.prediction_only || (!pl.residual && pl.picture_weight_1 >= 0 && pl.picture_weight_2 >= 0
            && pl.picture_weight_1 + pl.picture_weight_2 <= (1 << pl.picture_weight_bits)),
        "obmc_batch: plane %d: prediction_only needs residual NULL and picture weights >= 0 that sum to at most 1 << bits "
        "(a prediction of 8 bits); other pictures take the residual form", p);
    SCHRO_HIP_REQUIRE (pl.picture_weight_bits >= 0 && pl.picture_weight_bits <= 6,
        "obmc_batch: picture_weight_bits %d unsupported", pl.picture_weight_bits);
    // bits 0 with a gain other than 1: the reference's edge-block ROUND_SHIFT is
    // 1 << (0 - 1), an undefined shift (schromotion8.c:391-397) -- nothing to be exact to
    SCHRO_HIP_REQUIRE (pl.picture_weight_bits > 0 || pl.picture_weight_1 + pl.picture_weight_2 == 1,
        "obmc_batch: picture_weight_bits 0 needs weights that sum to 1");
    ObmcJob & j = all[p];
    memset (&j, 0, sizeof (j));
    const int hs = pl.component ? pl.chroma_h_shift : 0, vs = pl.component ? pl.chroma_v_shift : 0;
    // schromotion8.c:730-764
    j.xbsep = pl.xbsep_luma >> hs;
    j.ybsep = pl.ybsep_luma >> vs;
    j.xblen = pl.xblen_luma >> hs;
    j.yblen = pl.yblen_luma >> vs;
    // schro_params_verify_block_params, schroparams.c:241-272
    SCHRO_HIP_REQUIRE (((pl.xblen_luma | pl.yblen_luma | pl.xbsep_luma | pl.ybsep_luma) & 3) == 0,
        "obmc_batch: plane %d: luma block sizes and separations must be multiples of 4", p);
    SCHRO_HIP_REQUIRE (j.xbsep > 0 && j.ybsep > 0 && j.xblen >= j.xbsep && j.yblen >= j.ybsep
        && j.xblen <= 2 * j.xbsep && j.yblen <= 2 * j.ybsep && j.xblen <= 64 && j.yblen <= 64,
        "obmc_batch: plane %d block geometry %dx%d sep %dx%d unsupported", p, j.xblen, j.yblen,
        j.xbsep, j.ybsep);
    j.xoff = (j.xblen - j.xbsep) / 2;
    j.yoff = (j.yblen - j.ybsep) / 2;
    SCHRO_HIP_REQUIRE (pl.width >= j.xblen && pl.height >= j.yblen,
        "obmc_batch: plane %d smaller than one block", p);
    j.nbx = pl.x_num_blocks;
    j.nby = pl.y_num_blocks;
    SCHRO_HIP_REQUIRE (j.nbx > 0 && j.nby > 0, "obmc_batch: plane %d has no blocks", p);
    // schromotion8.c:794-797
    j.max_x_blocks = std::min (j.nbx - 1, (pl.width - j.xoff) / j.xbsep);
    j.max_y_blocks = std::min (j.nby - 1, (pl.height - j.yoff) / j.ybsep);
    j.mv_shift_x = hs;
    j.mv_shift_y = vs;
    j.prec = pl.mv_precision;
    j.wbits = pl.picture_weight_bits;
    j.w1 = pl.picture_weight_1;
    j.w2 = pl.picture_weight_2;
    j.comp = pl.component;
    j.mvs = (const uint8_t *) pl.mvs;
    j.ref[0] = pl.ref1;
    j.ref_stride[0] = pl.ref1_stride;
    j.ref[1] = pl.ref2 ? pl.ref2 : pl.ref1;
    j.ref_stride[1] = pl.ref2 ? pl.ref2_stride : pl.ref1_stride;
    j.residual = pl.residual;
    j.residual_stride = pl.residual ? pl.residual_stride : 0;
    j.res_bpp = pl.residual ? pl.residual_bpp : 2;
    j.out = pl.out;
    j.out_stride = pl.out_stride;
    j.w = pl.width;
    j.h = pl.height;
    // pair images (r04): the component is byte component - 1 of the (U, V) samples
    SCHRO_HIP_REQUIRE (!pl.ref_pair || (pl.mv_precision >= 1 && pl.component >= 1),
        "obmc_batch: plane %d: ref_pair is for the chroma components of half-pel references", p);
    j.ref_ps = pl.ref_pair ? 1 : 0;
    j.ref_cb = pl.ref_pair ? pl.component - 1 : 0;
    if (pl.mv_precision >= 1)   // the tiled half-pel layout: a plain plane or an image of another layout read as one goes out of bounds
      for (int r = 0; r < 2; r++)
        SCHRO_HIP_REQUIRE (j.ref_stride[r] % 512 == 0 && j.ref_stride[r] >= hp_chunks (j.w, j.ref_ps) * 512
            && ((uintptr_t) j.ref[r] & 127) == 0,
            "obmc_batch: plane %d: reference %d is not a half-pel image of this component (128-byte aligned, stride from "
            "schro_hip_upsampled_bytes / _pair_bytes)", p, r + 1);
    const int variant = variant_of (pl);
    const int nd_row = (variant == 1 && use_row) ? obmc_row_nd (j, false) : 0;
    // (a launch per row length: the kernels differ in registers and so in workgroups per CU)
    key[p] = pl.mv_precision | (variant << 4) | (nd_row << 8) | (nd_row ? 1 << 16 : 0) | (pl.prediction_only ? 1 << 19 : 0);
    row_nd[p] = nd_row;
  }
  // row kernel: the U and V planes of a picture (same vectors, blocks and sample windows) become
  // ONE job whose tile workgroups decode the blocks once; such pairs form their own launch
  // -- unless the batch is so small that the planes' tiles, one workgroup each, still fit the
  // device's slots at once (six per CU): then a pair's workgroup only runs twice as long (one
  // 2160p picture: 510 pair tiles against 1536 slots)
  long pair_tiles = 0;
  for (int p = 0; p < nplanes; p++)
    if (row_nd[p] && all[p].comp != 0)
      pair_tiles += (long) ((all[p].w + 127) / 128) * ((all[p].h + 31) / 32);
  const bool pairs_pay = obmc_row_merge_mode () == 2 || (obmc_row_merge_mode () == 1 && pair_tiles > 6L * ctx->cus);
  auto same_blocks = [](const ObmcJob & a, const ObmcJob & b) {
    return a.comp == 1 && b.comp == 2
        && a.mvs == b.mvs && a.w == b.w && a.h == b.h && a.nbx == b.nbx && a.nby == b.nby && a.xblen == b.xblen
        && a.yblen == b.yblen && a.xbsep == b.xbsep && a.ybsep == b.ybsep && a.mv_shift_x == b.mv_shift_x
        && a.mv_shift_y == b.mv_shift_y && a.res_bpp == b.res_bpp && a.ref_stride[0] == b.ref_stride[0]
        && a.ref_stride[1] == b.ref_stride[1] && a.prec == b.prec;
  };
  for (int p = 0; p + 1 < nplanes; p++) {
    const ObmcJob & a = all[p], &b = all[p + 1];
    // r04: the U and V planes of a picture from PAIR images: one job, one fetch per tap for both
    // (obmc_row.hip, UV form); what it does not take (eighth pel, other weights, long rows) reads
    // its component out of the pair images in obmc.hip
    if (a.ref_ps && b.ref_ps && use_row && variant_of (planes[p]) == 1 && variant_of (planes[p + 1]) == 1 && same_blocks (a, b)
        && a.ref[0] == b.ref[0] && a.ref[1] == b.ref[1] && !planes[p].prediction_only == !planes[p + 1].prediction_only) {
      const int nd = obmc_row_nd (a, true);
      if (nd) {
        row_nd[p] = row_nd[p + 1] = nd;
        key[p] = key[p + 1] = a.prec | (1 << 4) | (nd << 8) | (1 << 16) | (1 << 18) | (planes[p].prediction_only ? 1 << 19 : 0);
        p++;
      }
      continue;
    }
    if (pairs_pay && row_nd[p] && row_nd[p + 1] && key[p] == key[p + 1] && same_blocks (a, b)) {
      key[p] |= 1 << 17;
      key[p + 1] |= 1 << 17;
      p++;
    }
  }
  // one launch per (precision class, kernel) group, keeping plane order
  std::vector < char >done (nplanes, 0);
  for (int first = 0; first < nplanes; first++) {
    if (done[first])
      continue;
    const int prec = planes[first].mv_precision;
    int nd = (key[first] >> 8) & 0xff;
    const bool row = (key[first] >> 16) & 1;
    if (row)
      for (int p = first; p < nplanes; p++)
        if (!done[p] && key[p] == key[first])
          nd = std::max (nd, row_nd[p]);
    const bool paired = (key[first] >> 17) & 1, uv = (key[first] >> 18) & 1, pred_only = (key[first] >> 19) & 1;
    uint32_t *overflow = nullptr;
    if (pred_only) {
      if (!ctx->dc_gave_up) {
        SCHRO_HIP_CHECK (hipHostMalloc ((void **) &ctx->dc_gave_up, 64, hipHostMallocDefault));
        memset (ctx->dc_gave_up, 0, 64);
      }
      // (one number and one ring word per prediction_only CALL: all its launches share them)
      if (!pred_epoch) {
        pred_epoch = ++ctx->pred_epoch;
        const int slot = (int) (pred_epoch % SchroHipContext::kOvfRing);
        // a word still raised by a batch nobody has asked about keeps naming THAT batch
        if (!((volatile uint32_t *) ctx->dc_gave_up)[4 + slot])
          ctx->ovf_epoch[slot] = pred_epoch;
      }
      overflow = ctx->dc_gave_up + 4 + pred_epoch % SchroHipContext::kOvfRing;
    }
    const int variant = uv ? 4 : nd ? 3 : ((key[first] >> 4) & 15);
    std::vector < ObmcJob > jobs;
    int tile_base = 0;
    for (int p = first; p < nplanes; p++) {
      if (done[p] || key[p] != key[first])
        continue;
      done[p] = 1;
      ObmcJob j = all[p];
      int tiles_y;
      obmc_tiles (variant, j.w, j.h, j.xoff, &j.tiles_x, &tiles_y);
      j.tile_base = tile_base;
      obmc_item_geometry (&j);
      j.stamps = obmc_stamp_buffer ();
      j.nplanes = 1;
      if ((paired || uv) && j.comp == 2) {
        // the V plane joins the U plane in front of it (checked above)
        ObmcJob & a = jobs.back ();
        a.nplanes = 2;
        a.comp_b = j.comp;
        a.ref_b[0] = j.ref[0];
        a.ref_b[1] = j.ref[1];
        a.residual_b = j.residual;
        a.out_b = j.out;
        a.residual_stride_b = j.residual_stride;
        a.out_stride_b = j.out_stride;
        continue;
      }
      tile_base += j.tiles_x * tiles_y;
      jobs.push_back (j);
    }
    void *d_jobs;
    int r = push_args (ctx, jobs.data (), sizeof (ObmcJob) * jobs.size (), &d_jobs);
    if (r)
      return r;
    const uint32_t *d_order;
    r = obmc_tile_order (ctx, jobs, variant, tile_base, &d_order);
    if (r)
      return r;
    if (g_stamps)               // scratch runs: the dump describes the last launch only
      (void) hipMemsetAsync (g_stamps, 0, 16384 * 16 * 8, ctx->stream);
    {
      ProfileScope ps (ctx, SCHRO_HIP_KERNEL_OBMC);
      r = row ? launch_obmc_row (ctx->stream, (const ObmcJob *) d_jobs, (int) jobs.size (), tile_base, nd, uv ? 3 : paired ? 2 : 1, d_order, overflow)
          : launch_obmc (ctx->stream, (const ObmcJob *) d_jobs, (int) jobs.size (), tile_base, prec, variant, d_order, overflow);
    }
    if (r)
      return r;
  }
  return 0;
}

}                               // extern "C"

extern "C" {

// r05: the number the LATEST prediction_only call of schro_hip_obmc_batch on this context was given (1, 2, ...; 0: none
// yet).  A later SCHRO_HIP_ENEEDS_RESIDUAL from a synchronising call names the batch whose prediction did not fit
// 8 bits by this number, so a host that pipelines pictures knows WHICH picture to repeat in the residual order.
unsigned int
schro_hip_obmc_prediction_epoch (SchroHipContext * ctx)
{
  return ctx ? ctx->pred_epoch : 0u;
}

}                               // extern "C"
