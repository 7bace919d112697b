// obmc_row.hip -- the row formulation of OBMC (obmc_row_body.h) on half- and quarter-pel references: the kernels
// bench.py's headline runs; the dispatch over the three reference kinds (obmc_row_plain.hip: full pel,
// obmc_row_eighth.hip: eighth pel) and the host's side of the formulation -- which planes it takes
// (obmc_row_form), a geometry's weight table, a tile's record.

#include "obmc_row_body.h"

namespace schro {
namespace {

// Waves per SIMD: as many workgroups per CU as fit.  r03 (luma, workgroups per CU by LDS padding ->
// ms per 8 x 2160p luma launch): 2 0.278, 3 0.211, 4 0.180, 5 0.162, 6 0.153, 7 0.148 -- a saturating
// curve: the launch is no longer waiting for anything in particular (compiled-out stages: everything
// but the passes 0.056, the passes' arithmetic 0.044, their loads 0.047, the LDS atomics 0.009 ms).
// The 12-pixel-row kernel takes 69 registers: seven waves per SIMD at 19.8 KB of LDS; the
// 6-pixel-row kernels (chroma planes on their own: 344 blocks, a 73-word accumulator pitch) run five
// or six workgroups per CU; the UV kernels are the 12-byte-row kernel on 64-pixel tiles.
// r05: the prediction-only (U, V) kernel of 6-pixel rows at EIGHT workgroups per CU as well: its tables were trimmed to
// 20 464 B (accumulator margin 5 and pitch 75, 180 blocks, 960 items: what 6 x 6 blocks every 4 pixels need) and the
// compiler keeps it to 78 SGPRs at this occupancy (12 of them parked in VGPR lanes).  OBMC per 8 x 2160p step 0.1662 -> 0.1639 ms.
SCHRO_ROW_KERNEL (obmc_row_kernel_2_1, 6, 2, 1)
SCHRO_ROW_KERNEL (obmc_row_kernel_2_2, 6, 2, 2)
SCHRO_ROW_KERNEL (obmc_row_kernel_3_1, 7, 3, 1)
SCHRO_ROW_KERNEL (obmc_row_kernel_3_2, 5, 3, 2)
SCHRO_ROW_KERNEL (obmc_row_kernel_4_1, 6, 4, 1)
SCHRO_ROW_KERNEL (obmc_row_kernel_4_2, 5, 4, 2)
SCHRO_ROW_KERNEL (obmc_row_kernel_uv_2, 5, 2, 1, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_uv_3, 7, 3, 1, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_uv_4, 6, 4, 1, true)
// prediction_only launches (NORES, r05).  Without the residual's eight registers the 12-pixel-row kernel takes 60
// VGPRs and -- compiled for eight waves -- 78 SGPRs: EIGHT workgroups per CU (LDS 8 x 19 776 B = 158 KB; a CU admits
// floor (800 / (ceil (sgpr / 16) 16 + 16)) 256-thread workgroups: 7 at 81 .. 96 SGPRs).  8 x 2160p, same box: OBMC
// 0.1706 -> 0.1674 ms per step.  The (U, V) kernel of 6-pixel rows too (r05, above).
SCHRO_ROW_KERNEL (obmc_row_kernel_p_2_1, 6, 2, 1, false, kRTH, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_p_3_1, 8, 3, 1, false, kRTH, true)
#ifdef SCHRO_HIP_EXPERIMENTS
// r06, measured and not kept (HISTORY 9): the items of a block laid out so that every four lanes are the four rows of ONE
// 128-byte line (obmc_row_body.h: PAD) -- SCHRO_HIP_OBMC_PAD=1 in the experiments library
SCHRO_ROW_KERNEL (obmc_row_kernel_p_3_1_pad, 8, 3, 1, false, kRTH, true, 1, 1, false, true)
#endif
SCHRO_ROW_KERNEL (obmc_row_kernel_p_4_1, 7, 4, 1, false, kRTH, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_p_uv_2, 5, 2, 1, true, kRTH, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_p_uv_3, 8, 3, 1, true, kRTH, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_p_uv_4, 7, 4, 1, true, kRTH, true)
// r06, two segments per block row: the 24 / 16 block set's luma planes (24 = 2 x 12 pixels) and its 12-sample chroma
// rows from pair images (2 x 6 (U, V) samples)
SCHRO_ROW_KERNEL (obmc_row_kernel_h2_3_1, 6, 3, 1, false, kRTH, false, 1, 2)
SCHRO_ROW_KERNEL (obmc_row_kernel_h2_uv_3, 6, 3, 1, true, kRTH, false, 1, 2)
SCHRO_ROW_KERNEL (obmc_row_kernel_p_h2_3_1, 7, 3, 1, false, kRTH, true, 1, 2)
SCHRO_ROW_KERNEL (obmc_row_kernel_p_h2_uv_3, 6, 3, 1, true, kRTH, true, 1, 2)
// ... and of 16: the 32 / 16 block set (luma 32 = 2 x 16 pixels, its 16-sample chroma rows 2 x 8 (U, V) samples)
SCHRO_ROW_KERNEL (obmc_row_kernel_h2_4_1, 5, 4, 1, false, kRTH, false, 1, 2)
SCHRO_ROW_KERNEL (obmc_row_kernel_h2_uv_4, 5, 4, 1, true, kRTH, false, 1, 2)
SCHRO_ROW_KERNEL (obmc_row_kernel_p_h2_4_1, 6, 4, 1, false, kRTH, true, 1, 2)
SCHRO_ROW_KERNEL (obmc_row_kernel_p_h2_uv_4, 6, 4, 1, true, kRTH, true, 1, 2)

// r06, picture weights other than 1, 1 / 2 (fades: non-negative, adding up to 1 << bits): the 12-pixel-row and (U, V) forms
// with a prediction-only twin each (the headline's block set) ...
SCHRO_ROW_KERNEL (obmc_row_kernel_w_3_1, 6, 3, 1, false, kRTH, false, 1, 1, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_w_uv_3, 6, 3, 1, true, kRTH, false, 1, 1, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_w_p_3_1, 7, 3, 1, false, kRTH, true, 1, 1, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_w_p_uv_3, 7, 3, 1, true, kRTH, true, 1, 1, true)
// ... and one kernel per other form (with the residual's registers: it serves the prediction-only launches of its form too)
SCHRO_ROW_KERNEL (obmc_row_kernel_w_2_1, 5, 2, 1, false, kRTH, false, 1, 1, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_w_4_1, 6, 4, 1, false, kRTH, false, 1, 1, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_w_uv_2, 5, 2, 1, true, kRTH, false, 1, 1, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_w_uv_4, 6, 4, 1, true, kRTH, false, 1, 1, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_w_h2_3_1, 6, 3, 1, false, kRTH, false, 1, 2, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_w_h2_uv_3, 6, 3, 1, true, kRTH, false, 1, 2, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_w_h2_4_1, 6, 4, 1, false, kRTH, false, 1, 2, true)
SCHRO_ROW_KERNEL (obmc_row_kernel_w_h2_uv_4, 6, 4, 1, true, kRTH, false, 1, 2, true)

}                               // namespace

// the kernel of a form: nd dwords per row and segment, np planes per job (3: (U, V) pairs from pair images), ns segments
// per block row; nores: a prediction_only launch.  NULL: the form has no kernel
RowKernel
obmc_row_kernel_half (int nd, int np, int ns, bool nores, bool weighted)
{
  if (weighted) {
    if (np != 1 && np != 3)
      return nullptr;           // (two planes per job: the planes run as jobs of their own)
    const bool uv = np == 3;
    if (ns == 2)
      return nd == 3 ? (uv ? obmc_row_kernel_w_h2_uv_3 : obmc_row_kernel_w_h2_3_1) : nd == 4 ? (uv ? obmc_row_kernel_w_h2_uv_4 : obmc_row_kernel_w_h2_4_1) : nullptr;
    if (nd == 3)
      return uv ? (nores ? obmc_row_kernel_w_p_uv_3 : obmc_row_kernel_w_uv_3) : (nores ? obmc_row_kernel_w_p_3_1 : obmc_row_kernel_w_3_1);
    return nd == 2 ? (uv ? obmc_row_kernel_w_uv_2 : obmc_row_kernel_w_2_1) : nd == 4 ? (uv ? obmc_row_kernel_w_uv_4 : obmc_row_kernel_w_4_1) : nullptr;
  }
  if (ns == 2) {
    if (nd == 3 && np == 1)
      return nores ? obmc_row_kernel_p_h2_3_1 : obmc_row_kernel_h2_3_1;
    if (nd == 3 && np == 3)
      return nores ? obmc_row_kernel_p_h2_uv_3 : obmc_row_kernel_h2_uv_3;
    if (nd == 4 && np == 1)
      return nores ? obmc_row_kernel_p_h2_4_1 : obmc_row_kernel_h2_4_1;
    if (nd == 4 && np == 3)
      return nores ? obmc_row_kernel_p_h2_uv_4 : obmc_row_kernel_h2_uv_4;
    return nullptr;
  }
#ifdef SCHRO_HIP_EXPERIMENTS
  static const bool pad = SCHRO_ENV ("SCHRO_HIP_OBMC_PAD") && atoi (SCHRO_ENV ("SCHRO_HIP_OBMC_PAD")) != 0;
  if (pad && nores && nd == 3 && np == 1)
    return obmc_row_kernel_p_3_1_pad;
#endif
  if (nores)
    switch (nd * 10 + np) {
      case 23: return obmc_row_kernel_p_uv_2;
      case 33: return obmc_row_kernel_p_uv_3;
      case 43: return obmc_row_kernel_p_uv_4;
      case 21: return obmc_row_kernel_p_2_1;
      case 31: return obmc_row_kernel_p_3_1;
      case 41: return obmc_row_kernel_p_4_1;
    }
  switch (nd * 10 + np) {
    case 23: return obmc_row_kernel_uv_2;
    case 33: return obmc_row_kernel_uv_3;
    case 43: return obmc_row_kernel_uv_4;
    case 21: return obmc_row_kernel_2_1;
    case 22: return obmc_row_kernel_2_2;
    case 31: return obmc_row_kernel_3_1;
    case 32: return obmc_row_kernel_3_2;
    case 41: return obmc_row_kernel_4_1;
    case 42: return obmc_row_kernel_4_2;
  }
  return nullptr;
}

RowKernel obmc_row_kernel_plain (int nd, int np, int ns, bool nores, bool weighted);         // obmc_row_plain.hip
RowKernel obmc_row_kernel_eighth (int nd, int np, int ns, bool nores, bool weighted);        // obmc_row_eighth.hip

static RowKernel
row_kernel (int rk, int nd, int np, int ns, bool nores, bool weighted)
{
  return rk == 0 ? obmc_row_kernel_plain (nd, np, ns, nores, weighted) : rk == 3 ? obmc_row_kernel_eighth (nd, np, ns, nores, weighted)
      : obmc_row_kernel_half (nd, np, ns, nores, weighted);
}

// the reference kind of a precision: 0 plain planes, 1 half-pel images read at half / quarter pel, 3 at eighth pel
int
obmc_row_kind (int prec)
{
  return prec == 0 ? 0 : prec == 3 ? 3 : 1;
}

// weighted: picture weights other than 1, 1 / 2
bool
obmc_row_has_kernel (int prec, int nd, int np, int ns, bool weighted)
{
  return row_kernel (obmc_row_kind (prec), nd, np, ns, false, weighted) != nullptr;
}

// np: planes per job (1, 2); 3: (U, V) pairs from pair images
// (r06, measured and not kept: PERSISTENT workgroups -- a grid of what the device holds at once, every workgroup looping
// over the tiles blockIdx.x + k gridDim.x.  One tile per workgroup 0.167 ms of OBMC per 8 x 2160p step, eight persistent
// workgroups per CU 0.192, seven 0.192: workgroups that all start together stay in step -- every CU decodes at once,
// gathers at once, stores at once --, where the dispatcher's own refill staggers them.  HISTORY 9.)
int
launch_obmc_row (hipStream_t stream, const ObmcJob * d_jobs, int njobs, int total_tiles, int prec, int nd, int ns, int np,
    const uint32_t * d_order, uint32_t * overflow, const uint32_t * d_wtabs, bool weighted)
{
  // a prediction_only launch (the caller passes its overflow word exactly then: every job's residual is NULL)
  const RowKernel k = row_kernel (obmc_row_kind (prec), nd, np, ns, overflow != nullptr, weighted);
  if (!k)
    return set_error (SCHRO_HIP_EINVAL, "obmc (row): precision %d, %d dwords per row x %d segments x %d planes unsupported", prec, nd, ns, np);
  // scratch runs: SCHRO_HIP_OBMC_LDS_PAD = bytes of unused dynamic LDS per workgroup (fewer workgroups per CU)
  static const int lds_pad = SCHRO_ENV ("SCHRO_HIP_OBMC_LDS_PAD") ? atoi (SCHRO_ENV ("SCHRO_HIP_OBMC_LDS_PAD")) : 0;
  SCHRO_LAUNCH (k, dim3 (total_tiles), dim3 (kRThreads), (size_t) lds_pad, stream, d_jobs, njobs, d_order, overflow, d_wtabs);
  hipError_t e = hipGetLastError ();
  if (e != hipSuccess)
    return set_error (SCHRO_HIP_EDEVICE, "obmc (row) launch: %s", hipGetErrorString (e));
  return 0;
}

// Prediction dwords per block row (and segment) the row kernels run this plane with, *ns = the segments of a block
// row; 0: not their case (blocks wider than 32 samples, unaligned references, origins beyond 16 bits, a form without
// a kernel) -> obmc.hip.
// uv: as the U plane of a (U, V) pair from pair images (rows of up to 8 samples = 16 bytes, 64-pixel tiles)
int
obmc_row_form (const ObmcJob & j, bool uv, int *ns)
{
  const int ps = uv ? 1 : 0;
  *ns = 1;
  // (pair images go with (U, V) jobs and planes with plane jobs -- except at full pel, where a (U, V) job reads the
  // two PLAIN planes of each reference, j.ref and j.ref_b, and interleaves their rows itself)
  if (j.ref_ps != (j.prec == 0 ? 0 : ps))
    return 0;
  if (j.prec < 0 || j.prec > 3 || (j.xblen & 1) || j.xblen < 2 || j.yblen > 32)
    return 0;
  // picture weights: 1, 1 / 2, or (r06) any that are not negative and add up to 1 << bits -- the fades, for which the
  // reference's interior and edge arithmetic agree (obmc_row_body.h: blend_weighted); a gain or a negative weight: obmc.hip
  if (j.w1 < 0 || j.w2 < 0 || j.wbits < 0 || j.wbits > 6 || j.w1 + j.w2 != (1 << j.wbits))
    return 0;
  // rows of up to 16 bytes are one run; 24-byte rows (the 24 / 16 and 24 / 12 block sets) two segments of 12, 32-byte rows
  // (32 / 16: what the reference's encoder makes of 1080p and larger pictures by default, schroengine.c:411-453) two of 16;
  // 20 and 28 bytes (block lengths the syntax allows and no preset uses, schroparams.c:255-270) two of 10 and 14 in the
  // same kernels -- a segment's weights beyond its length are zero (obmc_row_weight_table)
  int seg_bytes = j.xblen << ps;
  if (seg_bytes > 16) {
    if (seg_bytes > 32 || (seg_bytes & 3))      // (halves of whole pixel pairs: luma lengths are multiples of 4)
      return 0;
    *ns = 2;
    seg_bytes /= 2;
  }
  // rim blocks keep their clamped fetch origins (get_block: at most (size + 32) << prec) in 16 bits
  if (((std::max (j.w, j.h) + 32) << j.prec) > 32767)
    return 0;
  if (j.prec == 0) {
    // plain planes: dword-aligned runs from a buffer of whole dwords
    if (((((uintptr_t) j.ref[0]) | ((uintptr_t) j.ref[1])) & 3) || ((j.ref_stride[0] | j.ref_stride[1]) & 3) || j.ref_stride[0] < j.w
        || j.ref_stride[1] < j.w || j.w < 4)
      return 0;
    if (uv && (!j.ref_b[0] || !j.ref_b[1] || ((((uintptr_t) j.ref_b[0]) | ((uintptr_t) j.ref_b[1])) & 3)))
      return 0;
  } else {
    if ((((uintptr_t) j.ref[0]) | ((uintptr_t) j.ref[1])) & 127)
      return 0;
    if (((j.ref_stride[0] | j.ref_stride[1]) & 511) || j.ref_stride[0] < hp_chunks (j.w, ps) * 512 || j.ref_stride[1] < hp_chunks (j.w, ps) * 512)
      return 0;
  }
  const int need = (seg_bytes + 3) / 4, nd = need <= 2 ? 2 : need;      // 2, 3 or 4
  if (!obmc_row_has_kernel (j.prec, nd, uv ? 3 : 1, *ns, j.w1 != 1 || j.wbits != 1))
    return 0;
  // the blocks (segments) that can meet a tile and their rows inside it fit the kernel's tables
  const int tw = uv ? RowGeo < 3, true >::kTW : RowGeo < 3, false >::kTW;
  // (a launch's order table names a tile in 16 bits -- plane_obmc.cpp: obmc_tile_order --: larger planes go to obmc.hip
  // instead of failing there)
  if ((long long) ((j.w + tw - 1) / tw) * ((j.h + kRTH - 1) / kRTH) > 0xffff)
    return 0;
  const int nbi = ((tw - 1 + j.xblen - 1) / j.xbsep + 1) * *ns, nbj = (kRTH - 1 + j.yblen - 1) / j.ybsep + 1;
  int blk_cap, item_cap;
  if (*ns == 2) {
    blk_cap = uv ? RowGeo < 3, true, 2 >::kBlk : RowGeo < 3, false, 2 >::kBlk;
    item_cap = uv ? RowGeo < 3, true, 2 >::kItem : RowGeo < 3, false, 2 >::kItem;
  } else if (nd == 4) {
    blk_cap = uv ? RowGeo < 4, true >::kBlk : RowGeo < 4, false >::kBlk;
    item_cap = uv ? RowGeo < 4, true >::kItem : RowGeo < 4, false >::kItem;
  } else {
    blk_cap = uv ? (nd <= 2 ? RowGeo < 2, true >::kBlk : RowGeo < 3, true >::kBlk) : (nd <= 2 ? RowGeo < 2, false >::kBlk : RowGeo < 3, false >::kBlk);
    item_cap = uv ? (nd <= 2 ? RowGeo < 2, true >::kItem : RowGeo < 3, true >::kItem)
        : (nd <= 2 ? RowGeo < 2, false >::kItem : RowGeo < 3, false >::kItem);
  }
  // the rows of one column of blocks (segments) inside a tile: the tile's rows, each under at most ceil (yblen / ybsep)
  // blocks -- or, for small overlaps the tighter count, the rows plus the overlaps of the block rows that meet it
  const int rows_col = std::min (kRTH * ((j.yblen + j.ybsep - 1) / j.ybsep), kRTH + (kRTH / j.ybsep + 2) * 2 * j.yoff);
  if (nbi > 255 || nbi * nbj > blk_cap || nbi * rows_col > item_cap)
    return 0;
  return nd;
}

// r05: the weight table a row kernel of `nd` dwords per row and `ns` segments copies into LDS for a job of this block
// geometry (RowGeo::kWTab words; the kernels made it themselves until r05, 2.2 k cycles of every tile's 25 k):
//   [0, 64 nd ns)    wx * wy of every (block row, pixel pair), two 16-bit products per word (<= 64 each); rows of
//                    2 * nd * ns words (segment after segment), zero beyond the block's width.  uv: a word per pixel,
//                    its weight for both components
//   [.., +4 xf)      folded x weights by edge type (4 x xf words, pairs as above; xf = 8, two segments: 2 nd ns)
//   [.. , +128)      folded y weights by edge type (4 x 32)
//   [.. , +wxn), +32 the two ramps, obmc_weight_1d (schromotion.c:40-69); wxn = 16, two segments: 32
// Folded: the 1-D weights of blocks that hang over the picture's rim (accumulate_slow's folding, schromotion8.c:673-693,
// by edge type instead of by pixel): the first block row / column folds its first 2 * offset weights, the last one
// everything from the block step on.
int
obmc_row_weight_words (int nd, int ns)
{
  const int wrow = 2 * nd * ns;
  return 32 * wrow + 4 * (ns == 1 ? 8 : wrow) + 128 + (ns == 1 ? 16 : 32) + 32;       // (= RowGeo < nd, *, ns >::kWTab)
}
static_assert (RowGeo < 3, false, 2 >::kWTab == 32 * 12 + 48 + 128 + 32 + 32 && RowGeo < 4, true, 2 >::kWTab == 32 * 16 + 64 + 128 + 32 + 32
    && RowGeo < 3, true >::kWTab == 64 * 3 + 32 + 128 + 16 + 32, "obmc_row_weight_words");

void
obmc_row_weight_table (const ObmcJob & j, int nd, int ns, bool uv, uint32_t * out)
{
  auto ramp = [](int x, int offset) { return offset == 1 ? (x == 0 ? 3 : 5) : 1 + (6 * x + offset - 1) / (2 * offset - 1); };
  auto weight = [&](int i, int blen, int offset) {
    if (offset == 0)
      return 8;
    if (i < 2 * offset)
      return ramp (i, offset);
    if (blen - 1 - i < 2 * offset)
      return ramp (blen - 1 - i, offset);
    return 8;
  };
  const int wrow = 2 * nd * ns, xf = ns == 1 ? 8 : wrow, wxn = ns == 1 ? 16 : 32;
  int wx[32] = { 0 }, wy[32] = { 0 };
  for (int i = 0; i < j.xblen && i < wxn; i++)
    wx[i] = weight (i, j.xblen, j.xoff);
  for (int i = 0; i < j.yblen && i < 32; i++)
    wy[i] = weight (i, j.yblen, j.yoff);
  auto folded = [](const int *w1, int idx, int blen, int bsep, int off, int type) {
    if (idx >= blen)
      return 0;
    int w = w1[idx];
    if ((type & 1) && idx < 2 * off)
      w += w1[2 * off - idx - 1];
    if ((type & 2) && idx >= bsep)
      w += w1[2 * (blen - off) - idx - 1];
    return w;
  };
  const int wcap = 32 * wrow;
  // word `pr` of a row -> the block's pixel (uv) or pixel pair it weights, -1: none.  One segment: word after word.  Two
  // segments: 2 nd words each (the kernels' seg_w), of which a segment of xblen / 2 pixels fills the first ones -- all of
  // them for rows of 24 and 32 bytes, 10 of 12 / 14 of 16 bytes for rows of 20 / 28
  const int seg_words = ns == 1 ? wrow : 2 * nd, seg_units = ns == 1 ? (uv ? j.xblen : j.xblen >> 1) : (uv ? j.xblen / 2 : j.xblen >> 2);
  auto unit_of = [&](int pr) {
    const int seg = pr / seg_words, q = pr - seg * seg_words;
    return q < seg_units ? seg * seg_units + q : -1;
  };
  for (int i = 0; i < wcap; i++) {
    const int r = i / wrow, u = unit_of (i - r * wrow);
    uint32_t v = 0;
    if (r < j.yblen && u >= 0) {
      if (uv)
        v = (uint32_t) (wx[u] * wy[r]) * 0x00010001u;
      else
        v = (uint32_t) (wx[2 * u] * wy[r]) | ((uint32_t) (wx[2 * u + 1] * wy[r]) << 16);
    }
    out[i] = v;
  }
  for (int t = 0; t < 4 * xf; t++) {
    const int type = t / xf, pr = t - type * xf;
    // (one segment: xf = 8 words, pixels beyond the block weigh nothing -- folded () --; two: xf = the row's words)
    const int u = ns == 1 ? pr : unit_of (pr);
    out[wcap + t] = u < 0 ? 0u : uv ? (uint32_t) folded (wx, u, j.xblen, j.xbsep, j.xoff, type) * 0x00010001u
        : (uint32_t) folded (wx, 2 * u, j.xblen, j.xbsep, j.xoff, type) | ((uint32_t) folded (wx, 2 * u + 1, j.xblen, j.xbsep, j.xoff, type) << 16);
  }
  const int fy = wcap + 4 * xf;
  for (int t = 0; t < 128; t++)
    out[fy + t] = (uint32_t) folded (wy, t & 31, j.yblen, j.ybsep, j.yoff, t >> 5);
  for (int i = 0; i < wxn; i++)
    out[fy + 128 + i] = (uint32_t) wx[i];
  for (int i = 0; i < 32; i++)
    out[fy + 128 + wxn + i] = (uint32_t) wy[i];
}

// r05: the record of tile (tx, ty) of a job in a row launch's order table (the kernels' decode set-up, done once per
// geometry on the host): x_lo | y_lo << 16, i_lo | j_lo << 16, nbi | nbj << 8 | ceil (2^16 / nbi) << 16 -- the blocks
// [i_lo, i_lo + nbi / ns) x [j_lo, j_lo + nbj) are the ones whose footprint meets the tile; nbi counts their SEGMENTS
void
obmc_row_tile_record (const ObmcJob & j, bool uv, int ns, int tx, int ty, uint32_t * rec)
{
  const int tw = uv ? RowGeo < 3, true >::kTW : RowGeo < 3, false >::kTW;
  const int x_lo = tx * tw, y_lo = ty * kRTH, x_hi = std::min (x_lo + tw, j.w), y_hi = std::min (y_lo + kRTH, j.h);
  const int i_lo = std::max (0, (x_lo + j.xoff - j.xblen + 2 * j.xbsep) / j.xbsep - 1);
  const int i_hi = std::min (j.nbx - 1, (x_hi - 1 + j.xoff) / j.xbsep);
  const int j_lo = std::max (0, (y_lo + j.yoff - j.yblen + 2 * j.ybsep) / j.ybsep - 1);
  const int j_hi = std::min (j.nby - 1, (y_hi - 1 + j.yoff) / j.ybsep);
  const int nbi = std::max (0, std::min (255, (i_hi - i_lo + 1) * ns)), nbj = std::max (0, std::min (255, j_hi - j_lo + 1));
  const uint32_t m16 = nbi > 1 ? (65536u + (uint32_t) nbi - 1u) / (uint32_t) nbi : 0u;
  rec[0] = (uint32_t) x_lo | ((uint32_t) y_lo << 16);
  rec[1] = (uint32_t) i_lo | ((uint32_t) j_lo << 16);
  rec[2] = (uint32_t) nbi | ((uint32_t) nbj << 8) | (m16 << 16);
}

int
obmc_row_tile_height ()
{
  return kRTH;
}

// the row kernels' tile width (obmc_tiles): 128 pixels, (U, V) pairs 64
int
obmc_row_tile_width (bool uv)
{
  return uv ? RowGeo < 3, true >::kTW : RowGeo < 3, false >::kTW;
}

}                               // namespace schro
