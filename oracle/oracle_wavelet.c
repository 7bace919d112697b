/*
 * oracle_wavelet.c -- CPU restatement of the Dirac integer lifting wavelets.
 * TEST INFRASTRUCTURE (see schro_oracle.h); never linked into the product.
 *
 * Follows schroedinger/schrowaveletorc.c (inverse :1460-2667, forward
 * :285-1458) and the kernel arithmetic of schroedinger/schroorc.orc as spelt
 * out in C by schroorc-dist.c under DISABLE_ORC.
 *
 * Formulation.  The reference runs each filter as a row-skewed in-place
 * schedule (e.g. schro_iiwt_desl_9_3, :1475-1538) or through line-cached
 * virtual frames (fidelity :1844, daub :1996).  In every case a value is read
 * in exactly the state the lifting order demands, so a 2-D inverse equals:
 * all vertical lifting steps (step k over all rows before step k+1), then per
 * row the horizontal lifting steps on the split halves, then the interleave
 * (+ output rounding shift).  Each filter is a short list of lifting steps on
 * two arrays A (even samples / low half) and B (odd samples / high half);
 * out-of-range neighbour indices clamp inside the SAME array, which is what
 * extend_N_M (:192-269) and the CLAMP(row, 0|1, h-2|h-1) rules produce.
 * oracle/ref_driver.c re-runs the reference's own skewed schedule on the
 * reference's compiled kernels to check this equivalence.
 */
#include "schro_oracle.h"
#include <stdlib.h>
#include <string.h>

#define ORACLE_RESTRICT __restrict__

enum {
  K_ADD2_22,                    /* (s0+s1+2)>>2, every add wraps       orc_add2_rshift_{add,sub}_*_22 (schroorc.orc:4-73) */
  K_AVG11,                      /* (s0+s1+1)>>1 without wrap           orc_add2_rshift_{add,sub}_*_11 (:76-134) */
  K_MAS4,                       /* (9(s1+s2)-(s0+s3)+rnd)>>sh          orc_mas4_*_1991 (:295-412) */
  K_HAAR_HALF,                  /* (s0+1)>>1 without wrap              orc_haar_synth_* avgsw t,0 (:985-1027) */
  K_HAAR_FULL,                  /* s0                                  orc_haar_synth_* addw */
  K_MAS8,                       /* (sum w[k] s[k] + rnd)>>8 in C int   mas8_add_s16 (schrowaveletorc.c:606-623) */
  K_MAS2                        /* ((s0+s1)*c+2048)>>12                orc_mas2_{add,sub}_* (:221-292) */
};

typedef struct {
  int target;                   /* 0: updates A from B, 1: updates B from A */
  int kind;
  int off;                      /* index of first neighbour, relative to i */
  int sign;                     /* +1 add, -1 subtract (inverse direction) */
  int c;                        /* MAS2 multiplier */
  int rnd;
  int sh;
  const int *taps;              /* MAS8 */
} Step;

static const int fid_s1[8] = { -2, 10, -25, 81, 81, -25, 10, -2 };
static const int fid_s2[8] = { 8, -21, 46, -161, -161, 46, -21, 8 };

/* Synthesis (inverse) step lists; forward = reverse order, opposite sign.
 * f0 schro_synth_ext_desl93 :1466, f1 schro_synth_ext_53 :1542,
 * f2 schro_synth_ext_135 :1616, f3/f4 haar :1697-1764,
 * f5 schro_synth_ext_fidelity :1768, f6 schro_synth_ext_daub97 :1894. */
static const Step steps_f0[] = {
  {0, K_ADD2_22, -1, -1, 0, 2, 2, 0},
  {1, K_MAS4, -1, +1, 0, 8, 4, 0},
};
static const Step steps_f1[] = {
  {0, K_ADD2_22, -1, -1, 0, 2, 2, 0},
  {1, K_AVG11, 0, +1, 0, 1, 1, 0},
};
static const Step steps_f2[] = {
  {0, K_MAS4, -2, -1, 0, 16, 5, 0},
  {1, K_MAS4, -1, +1, 0, 8, 4, 0},
};
static const Step steps_haar[] = {
  {0, K_HAAR_HALF, 0, -1, 0, 1, 1, 0},
  {1, K_HAAR_FULL, 0, +1, 0, 0, 0, 0},
};
static const Step steps_f5[] = {
  {1, K_MAS8, -3, +1, 0, 128, 8, fid_s1},
  {0, K_MAS8, -4, +1, 0, 127, 8, fid_s2},
};
static const Step steps_f6[] = {
  {0, K_MAS2, -1, -1, 1817, 2048, 12, 0},
  {1, K_MAS2, 0, -1, 3616, 2048, 12, 0},
  {0, K_MAS2, -1, +1, 217, 2048, 12, 0},
  {1, K_MAS2, 0, +1, 6497, 2048, 12, 0},
};

typedef struct {
  const Step *steps;
  int nsteps;
  int shift;                    /* 0 none, 1 wrapping (x+1)>>1, 2 non-wrapping avg(x,0) */
} Filter;

static const Filter filters[7] = {
  {steps_f0, 2, 1},
  {steps_f1, 2, 1},
  {steps_f2, 2, 1},
  {steps_haar, 2, 0},
  {steps_haar, 2, 2},
  {steps_f5, 2, 0},
  {steps_f6, 4, 1},
};

static int
ntaps (int kind)
{
  switch (kind) {
    case K_ADD2_22:
    case K_AVG11:
    case K_MAS2:
      return 2;
    case K_MAS4:
      return 4;
    case K_MAS8:
      return 8;
    default:
      return 1;
  }
}

static inline int
clampi (int x, int lo, int hi)
{
  return x < lo ? lo : (x > hi ? hi : x);
}

/* ------------------------------------------------------------------------ */
#define T int16_t
#define SUF s16
#define WRAP(x) ((int16_t)(x))
#define WMUL(a,b) ((int32_t)(a) * (int32_t)(b))     /* mulswl: exact 32-bit product */
#define WADD32(a,b) ((int32_t)(a) + (int32_t)(b))   /* cannot overflow for s16 inputs */
#define AVG(a,b) (((int32_t)(a) + (int32_t)(b) + 1) >> 1)
#include "oracle_wavelet_tmpl.h"
#undef T
#undef SUF
#undef WRAP
#undef WMUL
#undef WADD32
#undef AVG

#define T int32_t
#define SUF s32
#define WRAP(x) ((int32_t)(uint32_t)(x))
#define WMUL(a,b) ((int32_t)((uint32_t)(a) * (uint32_t)(b)))    /* mulll: low 32 bits */
#define WADD32(a,b) ((int32_t)((uint32_t)(a) + (uint32_t)(b)))
#define AVG(a,b) ((int32_t)(((int64_t)(a) + (int64_t)(b) + 1) >> 1))
#include "oracle_wavelet_tmpl.h"
#undef T
#undef SUF
#undef WRAP
#undef WMUL
#undef WADD32
#undef AVG

/* ------------------------------------------------------------------------ */

static int
check_args (void *data, int stride, int width, int height, int filter,
    int bpp)
{
  if (!data || width < 2 || height < 2 || (width & 1) || (height & 1))
    return -1;
  if (filter < 0 || filter > 6)
    return -1;
  if (bpp != 2 && bpp != 4)
    return -1;
  if (stride < width * bpp)
    return -1;
  return 0;
}

int
oracle_iiwt_2d (void *data, int stride, int width, int height, int filter,
    int bpp)
{
  if (check_args (data, stride, width, height, filter, bpp))
    return -1;
  if (bpp == 2)
    iiwt_2d_s16 ((int16_t *) data, stride, width, height, &filters[filter]);
  else
    iiwt_2d_s32 ((int32_t *) data, stride, width, height, &filters[filter]);
  return 0;
}

int
oracle_iwt_2d (void *data, int stride, int width, int height, int filter,
    int bpp)
{
  if (check_args (data, stride, width, height, filter, bpp))
    return -1;
  if (bpp == 2)
    iwt_2d_s16 ((int16_t *) data, stride, width, height, &filters[filter]);
  else
    iwt_2d_s32 ((int32_t *) data, stride, width, height, &filters[filter]);
  return 0;
}

/* schro_decoder_inverse_iwt_transform, schrodecoder.c:1831-1848 */
int
oracle_inverse_iwt_component (void *data, int stride, int iwt_width,
    int iwt_height, int depth, int filter, int bpp)
{
  int level;
  for (level = depth - 1; level >= 0; level--) {
    if (oracle_iiwt_2d (data, stride << level, iwt_width >> level,
            iwt_height >> level, filter, bpp))
      return -1;
  }
  return 0;
}

/* schro_frame_iwt_transform, schroencoder.c (level 0 first) */
int
oracle_forward_iwt_component (void *data, int stride, int iwt_width,
    int iwt_height, int depth, int filter, int bpp)
{
  int level;
  for (level = 0; level < depth; level++) {
    if (oracle_iwt_2d (data, stride << level, iwt_width >> level,
            iwt_height >> level, filter, bpp))
      return -1;
  }
  return 0;
}
