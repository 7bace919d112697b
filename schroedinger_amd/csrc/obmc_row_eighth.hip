// obmc_row_eighth.hip -- the row formulation of OBMC (obmc_row_body.h, RK 3) at eighth pel: the four taps of the tiled
// half-pel images blended with orc_combine4_nxm_u8's general weights (schroframe.c:2288-2413,
// schroorc.orc:1635-1662) on pairs of 16-bit sums.

#include "obmc_row_body.h"

namespace schro {
namespace {

SCHRO_ROW_KERNEL (obmc_row_eighth_2_1, 6, 2, 1, false, kRTH, false, 3)
SCHRO_ROW_KERNEL (obmc_row_eighth_3_1, 6, 3, 1, false, kRTH, false, 3)
SCHRO_ROW_KERNEL (obmc_row_eighth_4_1, 5, 4, 1, false, kRTH, false, 3)
SCHRO_ROW_KERNEL (obmc_row_eighth_uv_2, 5, 2, 1, true, kRTH, false, 3)
SCHRO_ROW_KERNEL (obmc_row_eighth_uv_3, 6, 3, 1, true, kRTH, false, 3)
SCHRO_ROW_KERNEL (obmc_row_eighth_uv_4, 5, 4, 1, true, kRTH, false, 3)
SCHRO_ROW_KERNEL (obmc_row_eighth_p_3_1, 7, 3, 1, false, kRTH, true, 3)
SCHRO_ROW_KERNEL (obmc_row_eighth_p_uv_3, 7, 3, 1, true, kRTH, true, 3)
SCHRO_ROW_KERNEL (obmc_row_eighth_h2_3_1, 6, 3, 1, false, kRTH, false, 3, 2)
SCHRO_ROW_KERNEL (obmc_row_eighth_h2_uv_3, 6, 3, 1, true, kRTH, false, 3, 2)
SCHRO_ROW_KERNEL (obmc_row_eighth_h2_4_1, 5, 4, 1, false, kRTH, false, 3, 2)
SCHRO_ROW_KERNEL (obmc_row_eighth_h2_uv_4, 5, 4, 1, true, kRTH, false, 3, 2)

// picture weights other than 1, 1 / 2 (fades)
SCHRO_ROW_KERNEL (obmc_row_eighth_w_3_1, 6, 3, 1, false, kRTH, false, 3, 1, true)
SCHRO_ROW_KERNEL (obmc_row_eighth_w_uv_3, 6, 3, 1, true, kRTH, false, 3, 1, true)
SCHRO_ROW_KERNEL (obmc_row_eighth_w_p_3_1, 6, 3, 1, false, kRTH, true, 3, 1, true)
SCHRO_ROW_KERNEL (obmc_row_eighth_w_p_uv_3, 6, 3, 1, true, kRTH, true, 3, 1, true)
// ... and one kernel per other form (it serves the prediction-only launches of its form too)
SCHRO_ROW_KERNEL (obmc_row_eighth_w_2_1, 5, 2, 1, false, kRTH, false, 3, 1, true)
SCHRO_ROW_KERNEL (obmc_row_eighth_w_4_1, 5, 4, 1, false, kRTH, false, 3, 1, true)
SCHRO_ROW_KERNEL (obmc_row_eighth_w_uv_2, 5, 2, 1, true, kRTH, false, 3, 1, true)
SCHRO_ROW_KERNEL (obmc_row_eighth_w_uv_4, 5, 4, 1, true, kRTH, false, 3, 1, true)
SCHRO_ROW_KERNEL (obmc_row_eighth_w_h2_3_1, 6, 3, 1, false, kRTH, false, 3, 2, true)
SCHRO_ROW_KERNEL (obmc_row_eighth_w_h2_uv_3, 6, 3, 1, true, kRTH, false, 3, 2, true)
SCHRO_ROW_KERNEL (obmc_row_eighth_w_h2_4_1, 5, 4, 1, false, kRTH, false, 3, 2, true)
SCHRO_ROW_KERNEL (obmc_row_eighth_w_h2_uv_4, 5, 4, 1, true, kRTH, false, 3, 2, true)

}                               // namespace

RowKernel
obmc_row_kernel_eighth (int nd, int np, int ns, bool nores, bool weighted)
{
  if (weighted) {
    if (np != 1 && np != 3)
      return nullptr;
    const bool uv = np == 3;
    if (ns == 2)
      return nd == 3 ? (uv ? obmc_row_eighth_w_h2_uv_3 : obmc_row_eighth_w_h2_3_1) : nd == 4 ? (uv ? obmc_row_eighth_w_h2_uv_4 : obmc_row_eighth_w_h2_4_1) : nullptr;
    if (nd == 3)
      return uv ? (nores ? obmc_row_eighth_w_p_uv_3 : obmc_row_eighth_w_uv_3) : (nores ? obmc_row_eighth_w_p_3_1 : obmc_row_eighth_w_3_1);
    return nd == 2 ? (uv ? obmc_row_eighth_w_uv_2 : obmc_row_eighth_w_2_1) : nd == 4 ? (uv ? obmc_row_eighth_w_uv_4 : obmc_row_eighth_w_4_1) : nullptr;
  }
  if (ns == 2)
    return nd == 3 && np == 1 ? obmc_row_eighth_h2_3_1 : nd == 3 && np == 3 ? obmc_row_eighth_h2_uv_3
        : nd == 4 && np == 1 ? obmc_row_eighth_h2_4_1 : nd == 4 && np == 3 ? obmc_row_eighth_h2_uv_4 : nullptr;
  if (nores && nd == 3 && np == 1)
    return obmc_row_eighth_p_3_1;
  if (nores && nd == 3 && np == 3)
    return obmc_row_eighth_p_uv_3;
  switch (nd * 10 + np) {
    case 21: return obmc_row_eighth_2_1;
    case 31: return obmc_row_eighth_3_1;
    case 41: return obmc_row_eighth_4_1;
    case 23: return obmc_row_eighth_uv_2;
    case 33: return obmc_row_eighth_uv_3;
    case 43: return obmc_row_eighth_uv_4;
  }
  return nullptr;
}

}                               // namespace schro
