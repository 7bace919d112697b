/*
 * ref_driver.c -- drives the REFERENCE's own compiled kernels
 * (oracle/_ref/libschroorc_ref.so = /root/reference/schroedinger/
 * schroorc-dist.c built unmodified with -DDISABLE_ORC) in the reference's
 * row-skewed, in-place schedule.  TEST INFRASTRUCTURE; only built where
 * /root/reference exists, only used by tests/test_oracle_vs_ref.py to check
 * that oracle_wavelet.c's "all vertical steps, then horizontal" formulation
 * is the same function as the reference's skewed schedule running on the
 * reference's arithmetic.
 *
 * Schedules restated from schroedinger/schrowaveletorc.c:
 *   f0 schro_iiwt_desl_9_3 :1475-1538 (look-ahead 7 / 3 rows)
 *   f1 schro_iiwt_5_3      :1551-1612 (2 / 1)
 *   f2 schro_iiwt_13_5     :1625-1693 (8 / 4)
 *   f3,f4 schro_iiwt_haar* :1697-1764 (1)
 *   f6 schro_iiwt_daub_9_7 :1996-2049 -- three chained line-cached virtual
 *      frames; driven here as three whole-frame passes on the same kernels
 *      (orc_mas2_*_op vertically, orc_mas2_*_ip horizontally)
 *   f5 has no Orc kernel in the reference (plain C in schrowaveletorc.c), so
 *      it is not driven here.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* prototypes as in schroorc-dist.c (ORC_RESTRICT dropped) */
void orc_add2_rshift_sub_s16_22_vert (int16_t * d1, const int16_t * s1, const int16_t * s2, int n);
void orc_add2_rshift_sub_s16_22 (int16_t * d1, const int16_t * s1, int n);
void orc_add2_rshift_add_s16_11_op (int16_t * d1, const int16_t * s1, const int16_t * s2, const int16_t * s3, int n);
void orc_add2_rshift_add_s16_11 (int16_t * d1, const int16_t * s1, int n);
void orc_mas4_vert_add_s16_1991 (int16_t * d1, const int16_t * s1, const int16_t * s2, const int16_t * s3, const int16_t * s4, int p1, int p2, int n);
void orc_mas4_vert_sub_s16_1991 (int16_t * d1, const int16_t * s1, const int16_t * s2, const int16_t * s3, const int16_t * s4, int p1, int p2, int n);
void orc_mas4_horiz_add_s16_1991_ip (int16_t * d1, const int16_t * s1, int p1, int p2, int n);
void orc_mas4_horiz_sub_s16_1991_ip (int16_t * d1, const int16_t * s1, int p1, int p2, int n);
void orc_mas2_add_s16_op (int16_t * d1, const int16_t * s1, const int16_t * s2, const int16_t * s3, int p1, int p2, int p3, int n);
void orc_mas2_sub_s16_op (int16_t * d1, const int16_t * s1, const int16_t * s2, const int16_t * s3, int p1, int p2, int p3, int n);
void orc_mas2_add_s16_ip (int16_t * d1, const int16_t * s1, int p1, int p2, int p3, int n);
void orc_mas2_sub_s16_ip (int16_t * d1, const int16_t * s1, int p1, int p2, int p3, int n);
void orc_haar_synth_s16 (int16_t * d1, int16_t * d2, int n);
void orc_haar_synth_int_s16 (int16_t * d1, const int16_t * s1, const int16_t * s2, int n);
void orc_haar_synth_rrshift1_int_s16 (int16_t * d1, const int16_t * s1, const int16_t * s2, int n);
void orc_interleave2_rrshift1_s16 (int16_t * d1, const int16_t * s1, const int16_t * s2, int n);
void orc_interleave2_s16 (int16_t * d1, const int16_t * s1, const int16_t * s2, int n);

void orc_add2_rshift_sub_s32_22_op (int32_t * d1, const int32_t * s1, const int32_t * s2, const int32_t * s3, int n);
void orc_add2_rshift_sub_s32_22 (int32_t * d1, const int32_t * s1, int n);
void orc_add2_rshift_add_s32_11_op (int32_t * d1, const int32_t * s1, const int32_t * s2, const int32_t * s3, int n);
void orc_add2_rshift_add_s32_11 (int32_t * d1, const int32_t * s1, int n);
void orc_mas4_vert_add_s32_1991_op (int32_t * d1, const int32_t * s0, const int32_t * s1, const int32_t * s2, const int32_t * s3, const int32_t * s4, int p1, int p2, int n);
void orc_mas4_vert_sub_s32_1991_op (int32_t * d1, const int32_t * s0, const int32_t * s1, const int32_t * s2, const int32_t * s3, const int32_t * s4, int p1, int p2, int n);
void orc_mas4_horiz_add_s32_1991_ip (int32_t * d1, const int32_t * s1, int p1, int p2, int n);
void orc_mas4_horiz_sub_s32_1991_ip (int32_t * d1, const int32_t * s1, int p1, int p2, int n);
void orc_mas2_add_s32_op (int32_t * d1, const int32_t * s1, const int32_t * s2, const int32_t * s3, int p1, int p2, int p3, int n);
void orc_mas2_sub_s32_op (int32_t * d1, const int32_t * s1, const int32_t * s2, const int32_t * s3, int p1, int p2, int p3, int n);
void orc_mas2_add_s32_ip (int32_t * d1, const int32_t * s1, int p1, int p2, int p3, int n);
void orc_mas2_sub_s32_ip (int32_t * d1, const int32_t * s1, int p1, int p2, int p3, int n);
void orc_haar_synth_s32 (int32_t * d1, int32_t * d2, int n);
void orc_haar_synth_int_s32 (int32_t * d1, const int32_t * s1, const int32_t * s2, int n);
void orc_haar_synth_rrshift1_int_s32 (int32_t * d1, const int32_t * s1, const int32_t * s2, int n);
void orc_interleave2_rrshift1_s32 (int32_t * d1, const int32_t * s1, const int32_t * s2, int n);
void orc_interleave2_s32 (int32_t * d1, const int32_t * s1, const int32_t * s2, int n);

static inline int
clampi (int x, int lo, int hi)
{
  return x < lo ? lo : (x > hi ? hi : x);
}

/* uniform names over the s16 (in-place d1) and s32 (_op) kernel flavours */
#define T int16_t
#define SUF s16
#define V_ADD2_SUB(d,a,b,n) orc_add2_rshift_sub_s16_22_vert (d, a, b, n)
#define V_AVG_ADD(d,a,b,n) orc_add2_rshift_add_s16_11_op (d, d, a, b, n)
#define V_MAS4_ADD(d,a,b,c,e,r,s,n) orc_mas4_vert_add_s16_1991 (d, a, b, c, e, r, s, n)
#define V_MAS4_SUB(d,a,b,c,e,r,s,n) orc_mas4_vert_sub_s16_1991 (d, a, b, c, e, r, s, n)
#define H_ADD2_SUB orc_add2_rshift_sub_s16_22
#define H_AVG_ADD orc_add2_rshift_add_s16_11
#define H_MAS4_ADD orc_mas4_horiz_add_s16_1991_ip
#define H_MAS4_SUB orc_mas4_horiz_sub_s16_1991_ip
#define MAS2_ADD_OP orc_mas2_add_s16_op
#define MAS2_SUB_OP orc_mas2_sub_s16_op
#define MAS2_ADD_IP orc_mas2_add_s16_ip
#define MAS2_SUB_IP orc_mas2_sub_s16_ip
#define HAAR_V orc_haar_synth_s16
#define HAAR_INT orc_haar_synth_int_s16
#define HAAR_INT_RR orc_haar_synth_rrshift1_int_s16
#define ILV_RR orc_interleave2_rrshift1_s16
#define ILV orc_interleave2_s16
#include "ref_driver_tmpl.h"
#undef T
#undef SUF
#undef V_ADD2_SUB
#undef V_AVG_ADD
#undef V_MAS4_ADD
#undef V_MAS4_SUB
#undef H_ADD2_SUB
#undef H_AVG_ADD
#undef H_MAS4_ADD
#undef H_MAS4_SUB
#undef MAS2_ADD_OP
#undef MAS2_SUB_OP
#undef MAS2_ADD_IP
#undef MAS2_SUB_IP
#undef HAAR_V
#undef HAAR_INT
#undef HAAR_INT_RR
#undef ILV_RR
#undef ILV

#define T int32_t
#define SUF s32
#define V_ADD2_SUB(d,a,b,n) orc_add2_rshift_sub_s32_22_op (d, d, a, b, n)
#define V_AVG_ADD(d,a,b,n) orc_add2_rshift_add_s32_11_op (d, d, a, b, n)
#define V_MAS4_ADD(d,a,b,c,e,r,s,n) orc_mas4_vert_add_s32_1991_op (d, d, a, b, c, e, r, s, n)
#define V_MAS4_SUB(d,a,b,c,e,r,s,n) orc_mas4_vert_sub_s32_1991_op (d, d, a, b, c, e, r, s, n)
#define H_ADD2_SUB orc_add2_rshift_sub_s32_22
#define H_AVG_ADD orc_add2_rshift_add_s32_11
#define H_MAS4_ADD orc_mas4_horiz_add_s32_1991_ip
#define H_MAS4_SUB orc_mas4_horiz_sub_s32_1991_ip
#define MAS2_ADD_OP orc_mas2_add_s32_op
#define MAS2_SUB_OP orc_mas2_sub_s32_op
#define MAS2_ADD_IP orc_mas2_add_s32_ip
#define MAS2_SUB_IP orc_mas2_sub_s32_ip
#define HAAR_V orc_haar_synth_s32
#define HAAR_INT orc_haar_synth_int_s32
#define HAAR_INT_RR orc_haar_synth_rrshift1_int_s32
#define ILV_RR orc_interleave2_rrshift1_s32
#define ILV orc_interleave2_s32
#include "ref_driver_tmpl.h"

/* One level, in place.  Returns -1 for filter 5 (no Orc kernels). */
int
refdrv_iiwt_2d (void *data, int stride, int width, int height, int filter,
    int bpp)
{
  if (filter == 5 || filter < 0 || filter > 6)
    return -1;
  if (bpp == 2)
    refdrv_iiwt_s16 ((int16_t *) data, stride, width, height, filter);
  else
    refdrv_iiwt_s32 ((int32_t *) data, stride, width, height, filter);
  return 0;
}
