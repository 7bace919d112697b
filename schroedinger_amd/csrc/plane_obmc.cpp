// plane_obmc.cpp -- plane layer: OBMC (schro_hip_obmc_batch: job geometry, kernel selection, the tile order, the
// prediction-only batches' numbering; obmc_row.hip, obmc.hip).

#include "schro_hip_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

using namespace schro;

static_assert (sizeof (ObmcJob) * kMaxJobs <= SchroHipContext::kArgSlotBytes, "a launch group of kMaxJobs OBMC jobs fits a table slot");

extern "C" {

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
  // where wave 0 of a workgroup ran (HW_ID: wave 0-3, simd 4-5)
  {
    unsigned long simd[4] = { 0, 0, 0, 0 };
    for (int b = 0; b < 16384; b++)
      if (h[b * 16 + 9])
        simd[(h[b * 16 + 14] >> 4) & 3]++;
    fprintf (stderr, "wave 0 of a workgroup on SIMD 0 / 1 / 2 / 3: %lu %lu %lu %lu\n", simd[0], simd[1], simd[2], simd[3]);
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
    const uint32_t ** d_order, int row_uv = -1, int row_ns = 1)     // row_uv >= 0: a row launch (1: of (U, V) pairs; row_ns: segments per block row) -- FOUR words per tile, obmc_row_tile_record
{
  *d_order = nullptr;
  static const bool enabled = !SCHRO_ENV ("SCHRO_HIP_OBMC_ORDER") || atoi (SCHRO_ENV ("SCHRO_HIP_OBMC_ORDER")) != 0;
  if (!enabled || variant < 1 || jobs.empty () || jobs.size () > 0xffff)
    return 0;
  const bool row = row_uv >= 0;
  const size_t words = row ? 4 : 1;     // per tile
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
    if (row) {                  // (what the tiles' records are made of)
      const ObmcJob & g = jobs[j];
      mix ((uint64_t) g.w | ((uint64_t) g.h << 20) | ((uint64_t) row_uv << 40) | (1ull << 41) | ((uint64_t) row_ns << 42));
      mix ((uint64_t) g.xbsep | ((uint64_t) g.ybsep << 8) | ((uint64_t) g.xblen << 16) | ((uint64_t) g.yblen << 24) | ((uint64_t) g.xoff << 32)
          | ((uint64_t) g.yoff << 40));
      mix ((uint64_t) g.nbx | ((uint64_t) g.nby << 20));
    }
  }
  constexpr int per_queue = SchroHipContext::kOrderSlots / SchroHipContext::kQueues;
  const int k0 = ctx->cur * per_queue;
  SchroHipContext::OrderSlot * slot = nullptr, *lru = &ctx->order_slots[k0];
  for (int k = k0; k < k0 + per_queue; k++) {
    SchroHipContext::OrderSlot & o = ctx->order_slots[k];
    if (o.d && o.hash == h && o.count == (size_t) total * words)
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
    if (slot->cap < keys.size () * words) {
      // grow-only; hipFree waits for the work that may still read the old table
      if (slot->d)
        SCHRO_HIP_CHECK (hipFree (slot->d));
      if (slot->h)
        SCHRO_HIP_CHECK (hipHostFree (slot->h));
      slot->d = nullptr;
      slot->h = nullptr;
      slot->cap = 0;
      const size_t cap = (keys.size () + keys.size () / 4) * words;
      SCHRO_HIP_CHECK (hipMalloc ((void **) &slot->d, cap * sizeof (uint32_t)));
      SCHRO_HIP_CHECK (hipHostMalloc ((void **) &slot->h, cap * sizeof (uint32_t), hipHostMallocDefault));
      slot->cap = cap;
    }
    if (!slot->copied)
      SCHRO_HIP_CHECK (hipEventCreateWithFlags (&slot->copied, hipEventDisableTiming));
    for (size_t k = 0; k < keys.size (); k++) {
      slot->h[k * words] = keys[k].entry;
      if (row) {
        const ObmcJob & g = jobs[keys[k].entry >> 16];
        const int t = (int) (keys[k].entry & 0xffffu);
        obmc_row_tile_record (g, row_uv != 0, row_ns, t % g.tiles_x, t / g.tiles_x, &slot->h[k * words + 1]);
      }
    }
    SCHRO_HIP_CHECK (hipMemcpyAsync (slot->d, slot->h, keys.size () * words * sizeof (uint32_t), hipMemcpyHostToDevice,
            ctx->stream));
    SCHRO_HIP_CHECK (hipEventRecord (slot->copied, ctx->stream));
    slot->copy_pending = true;
    slot->hash = h;
    slot->count = keys.size () * words;
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
  // default weights and blocks up to 16 wide (r06: or 24, as two segments), references of any precision (r06: plain
  // planes at full pel, eighth pel): the row kernels (obmc_row*.hip); SCHRO_HIP_OBMC_KERNEL=item sends them to obmc.hip's
  // item kernel (A/B runs: the second formulation the parity tests compare)
  static const bool use_row = !SCHRO_ENV ("SCHRO_HIP_OBMC_KERNEL") || strcmp (SCHRO_ENV ("SCHRO_HIP_OBMC_KERNEL"), "row") == 0;
  std::vector < ObmcJob > all (nplanes);
  std::vector < int >key (nplanes), row_nd (nplanes), row_ns (nplanes, 1);
  uint32_t pred_epoch = 0;      // (r05: this call's number among the context's prediction_only calls, once it has one)
  // r06: however the call ends, its ring word's event goes onto the queue behind whatever it has launched
  struct OvfDone {
    SchroHipContext *c = nullptr;
    int k = 0;
    void arm (SchroHipContext * ctx_, int k_) { c = ctx_; k = k_; }
    ~OvfDone () { if (c && c->ovf_ev[k]) (void) hipEventRecord (c->ovf_ev[k], c->stream); }
  } ovf_done;
  for (int p = 0; p < nplanes; p++) {
    const SchroHipObmcPlane & pl = planes[p];
    // (residual NULL: nothing to add -- the prediction alone, clamped)
    SCHRO_HIP_REQUIRE (pl.mvs && pl.ref1 && pl.out, "obmc_batch: plane %d has a NULL pointer", p);
    SCHRO_HIP_REQUIRE (pl.mv_precision >= 0 && pl.mv_precision <= 3,
        "obmc_batch: mv_precision %d out of range", pl.mv_precision);
    SCHRO_HIP_REQUIRE (pl.component >= 0 && pl.component <= 2, "obmc_batch: bad component");
    SCHRO_HIP_REQUIRE (!pl.residual || pl.residual_bpp == 2 || pl.residual_bpp == 4, "obmc_batch: residual bpp");
    // r04: the prediction alone, for the wavelet's combine form: it must fit the u8 plane it is written to
    SCHRO_HIP_REQUIRE (pl.prediction_only >= 0 && pl.prediction_only <= 2, "obmc_batch: plane %d: prediction_only is 0, 1 or 2", p);
    // r06, prediction_only 2: the prediction - 128 into an s16 plane (schro_motion_render's add = FALSE / schro_motion_render_cuda's
    // dest): 16-bit arithmetic carries whatever the weights and DC values give
    SCHRO_HIP_REQUIRE (pl.prediction_only != 2 || (!pl.residual && ((uintptr_t) pl.out | (uintptr_t) pl.out_stride) % 2 == 0),
        "obmc_batch: plane %d: prediction_only 2 needs residual NULL and an s16 plane in `out`", p);
    SCHRO_HIP_REQUIRE (pl.prediction_only != 1 || (!pl.residual && pl.picture_weight_1 >= 0 && pl.picture_weight_2 >= 0
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
    j.out_s16 = pl.prediction_only == 2;
    const int variant = variant_of (pl);
    // (obmc_row_form looks at the weights itself: 1, 1 / 2 and, r06, every non-negative pair that adds up to 1 << bits)
    const int nd_row = (use_row && !j.out_s16) ? obmc_row_form (j, false, &row_ns[p]) : 0;
    // (a launch per row length: the kernels differ in registers and so in workgroups per CU)
    key[p] = pl.mv_precision | (variant << 4) | (nd_row << 8) | (nd_row ? 1 << 16 : 0) | (pl.prediction_only == 1 ? 1 << 19 : 0)
        | (nd_row ? row_ns[p] << 20 : 0) | (j.out_s16 ? 1 << 22 : 0);
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
        && a.ref_stride[1] == b.ref_stride[1] && a.prec == b.prec && a.w1 == b.w1 && a.w2 == b.w2 && a.wbits == b.wbits;
  };
  for (int p = 0; p + 1 < nplanes; p++) {
    const ObmcJob & a = all[p], &b = all[p + 1];
    // r04: the U and V planes of a picture from PAIR images: one job, one fetch per tap for both
    // (obmc_row.hip, UV form); what it does not take (eighth pel, other weights, long rows) reads
    // its component out of the pair images in obmc.hip
    // r06: ... and at full pel the U and V planes of a picture (plain planes: two per reference) the same way -- the job
    // carries the V planes in ref_b and the kernel interleaves the rows
    const bool uv_plain = a.prec == 0 && b.prec == 0 && !a.ref_ps && !b.ref_ps;
    if (((a.ref_ps && b.ref_ps && a.ref[0] == b.ref[0] && a.ref[1] == b.ref[1]) || uv_plain)
        && use_row && same_blocks (a, b)
        && planes[p].prediction_only == planes[p + 1].prediction_only && !a.out_s16 && !b.out_s16) {
      int ns;
      ObmcJob au = a;
      au.ref_b[0] = b.ref[0];
      au.ref_b[1] = b.ref[1];
      const int nd = obmc_row_form (au, true, &ns);
      if (nd) {
        row_nd[p] = row_nd[p + 1] = nd;
        row_ns[p] = row_ns[p + 1] = ns;
        key[p] = key[p + 1] = a.prec | (1 << 4) | (nd << 8) | (1 << 16) | (1 << 18) | (planes[p].prediction_only == 1 ? 1 << 19 : 0) | (ns << 20);
        p++;
        continue;
      }
      if (!uv_plain)            // (pair images outside the row kernels' case: obmc.hip reads the components out of them)
        continue;
    }
    if (pairs_pay && row_nd[p] && row_nd[p + 1] && key[p] == key[p + 1] && same_blocks (a, b)
        && obmc_row_has_kernel (a.prec, row_nd[p], 2, row_ns[p], a.w1 != 1 || a.wbits != 1)) {
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
    const int ns = row ? std::max (1, (key[first] >> 20) & 3) : 1;
    // (two planes per job: every row length of the group needs the kernel)
    // (a launch group's planes have one kind of weights: the key carries the variant)
    const bool weighted = row && (all[first].w1 != 1 || all[first].wbits != 1);
    SCHRO_HIP_REQUIRE (!row || obmc_row_has_kernel (prec, nd, uv ? 3 : paired ? 2 : 1, ns, weighted),
        "obmc_batch: no row kernel for precision %d, %d dwords x %d segments per row, %s", prec, nd, ns, uv ? "(U, V) pairs" : paired ? "two planes per job" : "one plane per job");
    uint32_t *overflow = nullptr;
    if (pred_only) {
      if (!ctx->dc_gave_up) {
        SCHRO_HIP_CHECK (hipHostMalloc ((void **) &ctx->dc_gave_up, 64, hipHostMallocDefault));
        memset (ctx->dc_gave_up, 0, 64);
      }
      // (one number and one ring word per prediction_only CALL: all its launches share them; r06: the word is this
      // batch's alone until the event behind its launches -- OvfDone below -- has fired and the word has been read)
      if (!pred_epoch) {
        pred_epoch = ++ctx->pred_epoch;
        if (pred_epoch == 0)      // (0 means "none")
          pred_epoch = ++ctx->pred_epoch;
        const int rc = pred_overflow_claim (ctx, pred_epoch);
        if (rc)
          return rc;
        ovf_done.arm (ctx, (int) (pred_epoch % SchroHipContext::kOvfRing));
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
    // r05, experiments build only (SCHRO_HIP_OBMC_STRIP=1): the 12 / 8 block set on one-byte planes by the strip kernel,
    // accumulator in registers (obmc_strip.hip: bit-exact, 2.5 x slower -- the gather's lines from L2 are the bound, not the
    // LDS tile); its waves take strips of 15 block columns x segments of 8 block rows
#ifdef SCHRO_HIP_EXPERIMENTS
    static const bool use_strip = SCHRO_ENV ("SCHRO_HIP_OBMC_STRIP") && atoi (SCHRO_ENV ("SCHRO_HIP_OBMC_STRIP")) != 0;
    bool strip = use_strip && row && !paired && !uv && nd == 3 && ns == 1;
    for (size_t k = 0; strip && k < jobs.size (); k++)
      strip = obmc_strip_ok (jobs[k]);
    if (strip) {
      const int seg_rows = 8;
      int wave_base = 0;
      for (ObmcJob & j : jobs) {
        int strips, segs;
        obmc_strip_tiles (j, seg_rows, &strips, &segs);
        j.tiles_x = strips;
        j.tile_base = wave_base;
        wave_base += strips * segs;
      }
      void *d_sjobs;
      int rs = push_args (ctx, jobs.data (), sizeof (ObmcJob) * jobs.size (), &d_sjobs);
      if (rs)
        return rs;
      ProfileScope ps (ctx, SCHRO_HIP_KERNEL_OBMC);
      rs = launch_obmc_strip (ctx->stream, (const ObmcJob *) d_sjobs, (int) jobs.size (), wave_base, seg_rows, pred_only, overflow, ctx->cus);
      if (rs)
        return rs;
      continue;
    }
#endif
    // r05: a row launch's weight tables, one per block geometry among its jobs (nearly always one), made here instead of
    // by every tile; the jobs name theirs in `ipw` (a field of the item kernels, which these launches do not use).  A
    // group with more geometries than a table slot holds (35 .. 48) goes out in several launches.
    std::vector < ObmcJob > all_jobs;
    all_jobs.swap (jobs);
    const int all_tiles = tile_base;
    int r = 0;
    for (size_t first_job = 0; first_job < all_jobs.size () && !r;) {
      void *d_wtabs = nullptr;
      size_t end_job = all_jobs.size ();
      std::vector < uint32_t > tabs;
      if (row) {
        const size_t words = (size_t) obmc_row_weight_words (nd, ns);
        std::vector < uint32_t > one (words);
        for (size_t n = first_job; n < end_job; n++) {
          ObmcJob & j = all_jobs[n];
          obmc_row_weight_table (j, nd, ns, uv, one.data ());
          size_t k = 0;
          while (k * words < tabs.size () && memcmp (&tabs[k * words], one.data (), words * 4))
            k++;
          if (k * words == tabs.size ()) {
            if ((k + 1) * words * 4 > SchroHipContext::kArgSlotBytes) {
              end_job = n;      // (this job starts the next launch)
              break;
            }
            tabs.insert (tabs.end (), one.begin (), one.end ());
          }
          j.ipw = (int) k;
        }
        r = push_args (ctx, tabs.data (), tabs.size () * 4, &d_wtabs);
        if (r)
          return r;
      }
      // this launch's jobs, their tiles counted from its first one
      jobs.assign (all_jobs.begin () + (long) first_job, all_jobs.begin () + (long) end_job);
      const int base = jobs.front ().tile_base;
      tile_base = (end_job < all_jobs.size () ? all_jobs[end_job].tile_base : all_tiles) - base;
      for (ObmcJob & j : jobs)
        j.tile_base -= base;
      first_job = end_job;
      void *d_jobs;
      r = push_args (ctx, jobs.data (), sizeof (ObmcJob) * jobs.size (), &d_jobs);
      if (r)
        return r;
      const uint32_t *d_order;
      r = obmc_tile_order (ctx, jobs, variant, tile_base, &d_order, row ? (uv ? 1 : 0) : -1, ns);
      if (r)
        return r;
#ifndef SCHRO_HIP_EXPERIMENTS
      // (the product's row kernels take their tiles from the table only; obmc_row_form admits no plane that would not
      // get one -- more than 65535 tiles --, so this is an internal inconsistency, not a property of the caller's planes)
      if (row && !d_order)
        return set_error (SCHRO_HIP_EINVAL, "obmc_batch: internal: no tile table for a row launch of %zu jobs", jobs.size ());
#endif
      if (g_stamps)             // scratch runs: the dump describes the last launch only
        (void) hipMemsetAsync (g_stamps, 0, 16384 * 16 * 8, ctx->stream);
      {
        ProfileScope ps (ctx, SCHRO_HIP_KERNEL_OBMC);
        r = row ? launch_obmc_row (ctx->stream, (const ObmcJob *) d_jobs, (int) jobs.size (), tile_base, prec, nd, ns, uv ? 3 : paired ? 2 : 1, d_order, overflow, (const uint32_t *) d_wtabs, weighted)
            : launch_obmc (ctx->stream, (const ObmcJob *) d_jobs, (int) jobs.size (), tile_base, prec, variant, d_order, overflow);
      }
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
