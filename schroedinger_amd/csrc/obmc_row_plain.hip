// obmc_row_plain.hip -- the row formulation of OBMC (obmc_row_body.h, RK 0) on PLAIN planar references: mv_precision 0,
// the reference encoder's default (schroencoder.c:4488) and its own test stream's (testsuite/test_stream.drc).  No
// upsample stage runs for such pictures (schrodecoder.c:1596-1601): the references are the decoder's pictures as
// they are, one plane per component, and a window row is one dword-aligned run of one plane.
//
// References: schromotion8.c:303-335 (get_block), schroframe.c:2111-2122 (prec 0), :1940-1998 (the aprons the
// reference's frames carry, here the edge class's clamps).

#include "obmc_row_body.h"

namespace schro {
namespace {

// (waves per SIMD: as obmc_row.hip's kernels of the same shape; one tap per window: fewer registers in flight)
SCHRO_ROW_KERNEL (obmc_row_plain_2_1, 6, 2, 1, false, kRTH, false, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_2_2, 6, 2, 2, false, kRTH, false, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_3_1, 7, 3, 1, false, kRTH, false, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_3_2, 5, 3, 2, false, kRTH, false, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_4_1, 7, 4, 1, false, kRTH, false, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_4_2, 5, 4, 2, false, kRTH, false, 0)
// prediction_only launches (the combine form): the U and V planes of a picture are two plain planes, one job
SCHRO_ROW_KERNEL (obmc_row_plain_p_2_1, 6, 2, 1, false, kRTH, true, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_p_2_2, 6, 2, 2, false, kRTH, true, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_p_3_1, 8, 3, 1, false, kRTH, true, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_p_4_1, 8, 4, 1, false, kRTH, true, 0)
// r06: the U and V planes of a picture as ONE (U, V) job -- the two plain planes' rows interleaved in registers, then the
// pair images' accumulator and finish (64-pixel tiles, one decode, one pass per item for both planes)
SCHRO_ROW_KERNEL (obmc_row_plain_uv_2, 5, 2, 1, true, kRTH, false, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_uv_3, 7, 3, 1, true, kRTH, false, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_uv_4, 7, 4, 1, true, kRTH, false, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_p_uv_2, 5, 2, 1, true, kRTH, true, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_p_uv_3, 8, 3, 1, true, kRTH, true, 0)
SCHRO_ROW_KERNEL (obmc_row_plain_p_uv_4, 7, 4, 1, true, kRTH, true, 0)
// the 24 / 16 block set's luma rows: two segments of 12
SCHRO_ROW_KERNEL (obmc_row_plain_h2_3_1, 6, 3, 1, false, kRTH, false, 0, 2)
SCHRO_ROW_KERNEL (obmc_row_plain_p_h2_3_1, 7, 3, 1, false, kRTH, true, 0, 2)
// ... and, as (U, V) jobs, the 12-sample chroma rows of the 24 / 12 set; the 32 / 16 set -- the reference encoder's default for
// 1080p and larger pictures, at its default mv_precision 0 -- as two segments of 16 pixels / of 8 (U, V) samples
SCHRO_ROW_KERNEL (obmc_row_plain_h2_uv_3, 6, 3, 1, true, kRTH, false, 0, 2)
SCHRO_ROW_KERNEL (obmc_row_plain_p_h2_uv_3, 6, 3, 1, true, kRTH, true, 0, 2)
SCHRO_ROW_KERNEL (obmc_row_plain_h2_4_1, 5, 4, 1, false, kRTH, false, 0, 2)
SCHRO_ROW_KERNEL (obmc_row_plain_h2_uv_4, 5, 4, 1, true, kRTH, false, 0, 2)
SCHRO_ROW_KERNEL (obmc_row_plain_p_h2_4_1, 6, 4, 1, false, kRTH, true, 0, 2)
SCHRO_ROW_KERNEL (obmc_row_plain_p_h2_uv_4, 6, 4, 1, true, kRTH, true, 0, 2)

// picture weights other than 1, 1 / 2 (fades)
SCHRO_ROW_KERNEL (obmc_row_plain_w_3_1, 6, 3, 1, false, kRTH, false, 0, 1, true)
SCHRO_ROW_KERNEL (obmc_row_plain_w_uv_3, 6, 3, 1, true, kRTH, false, 0, 1, true)
SCHRO_ROW_KERNEL (obmc_row_plain_w_p_3_1, 7, 3, 1, false, kRTH, true, 0, 1, true)
SCHRO_ROW_KERNEL (obmc_row_plain_w_p_uv_3, 7, 3, 1, true, kRTH, true, 0, 1, true)
// ... and one kernel per other form (it serves the prediction-only launches of its form too)
SCHRO_ROW_KERNEL (obmc_row_plain_w_2_1, 6, 2, 1, false, kRTH, false, 0, 1, true)
SCHRO_ROW_KERNEL (obmc_row_plain_w_4_1, 6, 4, 1, false, kRTH, false, 0, 1, true)
SCHRO_ROW_KERNEL (obmc_row_plain_w_uv_2, 5, 2, 1, true, kRTH, false, 0, 1, true)
SCHRO_ROW_KERNEL (obmc_row_plain_w_uv_4, 6, 4, 1, true, kRTH, false, 0, 1, true)
SCHRO_ROW_KERNEL (obmc_row_plain_w_h2_3_1, 6, 3, 1, false, kRTH, false, 0, 2, true)
SCHRO_ROW_KERNEL (obmc_row_plain_w_h2_uv_3, 6, 3, 1, true, kRTH, false, 0, 2, true)
SCHRO_ROW_KERNEL (obmc_row_plain_w_h2_4_1, 6, 4, 1, false, kRTH, false, 0, 2, true)
SCHRO_ROW_KERNEL (obmc_row_plain_w_h2_uv_4, 6, 4, 1, true, kRTH, false, 0, 2, true)

}                               // namespace

RowKernel
obmc_row_kernel_plain (int nd, int np, int ns, bool nores, bool weighted)
{
  if (weighted) {
    if (np != 1 && np != 3)
      return nullptr;
    const bool uv = np == 3;
    if (ns == 2)
      return nd == 3 ? (uv ? obmc_row_plain_w_h2_uv_3 : obmc_row_plain_w_h2_3_1) : nd == 4 ? (uv ? obmc_row_plain_w_h2_uv_4 : obmc_row_plain_w_h2_4_1) : nullptr;
    if (nd == 3)
      return uv ? (nores ? obmc_row_plain_w_p_uv_3 : obmc_row_plain_w_uv_3) : (nores ? obmc_row_plain_w_p_3_1 : obmc_row_plain_w_3_1);
    return nd == 2 ? (uv ? obmc_row_plain_w_uv_2 : obmc_row_plain_w_2_1) : nd == 4 ? (uv ? obmc_row_plain_w_uv_4 : obmc_row_plain_w_4_1) : nullptr;
  }
  if (ns == 2) {
    if (nd == 3)
      return np == 1 ? (nores ? obmc_row_plain_p_h2_3_1 : obmc_row_plain_h2_3_1) : np == 3 ? (nores ? obmc_row_plain_p_h2_uv_3 : obmc_row_plain_h2_uv_3) : nullptr;
    if (nd == 4)
      return np == 1 ? (nores ? obmc_row_plain_p_h2_4_1 : obmc_row_plain_h2_4_1) : np == 3 ? (nores ? obmc_row_plain_p_h2_uv_4 : obmc_row_plain_h2_uv_4) : nullptr;
    return nullptr;
  }
  if (nores)
    switch (nd * 10 + np) {
      case 23: return obmc_row_plain_p_uv_2;
      case 33: return obmc_row_plain_p_uv_3;
      case 43: return obmc_row_plain_p_uv_4;
      case 21: return obmc_row_plain_p_2_1;
      case 22: return obmc_row_plain_p_2_2;
      case 31: return obmc_row_plain_p_3_1;
      case 41: return obmc_row_plain_p_4_1;
    }
  switch (nd * 10 + np) {
    case 23: return obmc_row_plain_uv_2;
    case 33: return obmc_row_plain_uv_3;
    case 43: return obmc_row_plain_uv_4;
    case 21: return obmc_row_plain_2_1;
    case 22: return obmc_row_plain_2_2;
    case 31: return obmc_row_plain_3_1;
    case 32: return obmc_row_plain_3_2;
    case 41: return obmc_row_plain_4_1;
    case 42: return obmc_row_plain_4_2;
  }
  return nullptr;
}

}                               // namespace schro
