/*
 * oracle_motion.c -- CPU restatement of the production OBMC renderer.
 * TEST INFRASTRUCTURE (see schro_oracle.h); never linked into the product.
 *
 * Follows schroedinger/schromotion8.c (schro_motion_render_u8 :700-929,
 * get_block :303-335, predict_block :542-568, predict_and_acc :570-657,
 * accumulate_slow :659-698, the five run-time Orc programs :15-167),
 * schroedinger/schromotion.c (get_ramp :40-49, init_obmc_weight :52-93) and
 * the sub-pel block fetch of schroedinger/schroframe.c
 * (get_block_fast_precN :2459-2482, _fast_prec3 :2288-2413, _fast_prec1
 * :2166-2185, subdata_prec0/1 :2111-2122,2187-2201).
 *
 * Deliberately kept in the reference's own shape -- block SCATTER into an s16
 * accumulator, interior blocks and edge blocks on different arithmetic,
 * block-position clamp, reads from materialised 32-pixel aprons -- because
 * the GPU path is a per-pixel GATHER with coordinate clamping; the parity
 * tests are what show the two are the same function.
 *
 * The five Orc programs have no C body in the reference; their opcodes are
 * restated with the semantics schroorc-dist.c gives the same opcodes:
 * convubw zero-extend, mullw low 16 bits, addw wrap, shrsw arithmetic,
 * avgub (a+b+1)>>1, convsuswb clamp to 0..255.
 */
#include "schro_oracle.h"
#include <stdlib.h>
#include <stddef.h>
#include <string.h>

#define MAX_BLK 64              /* SCHRO_LIMIT_BLOCK_SIZE, schrolimits.h:67; get_block clamps the block origin so every read stays inside the 32-pixel apron */

static inline int
clampi (int x, int lo, int hi)
{
  return x < lo ? lo : (x > hi ? hi : x);
}

/* schromotion.c:40-49 */
static int
get_ramp (int x, int offset)
{
  if (offset == 1) {
    if (x == 0)
      return 3;
    return 5;
  }
  return 1 + (6 * x + offset - 1) / (2 * offset - 1);
}

/* schromotion.c:52-93 (1-D part) */
static void
init_weights (int *w, int blen, int offset)
{
  int i;
  for (i = 0; i < blen; i++) {
    if (offset == 0)
      w[i] = 8;
    else if (i < 2 * offset)
      w[i] = get_ramp (i, offset);
    else if (blen - 1 - i < 2 * offset)
      w[i] = get_ramp (blen - 1 - i, offset);
    else
      w[i] = 8;
  }
}

typedef struct {
  const OracleMotionVector *mvs;
  const OracleMotionParams *p;
  int k;
  const OracleUpComp *ref[2];
  int xbsep, ybsep, xblen, yblen, xoffset, yoffset;
  int width, height;
  int max_fast_x, max_fast_y;
  int prec, bits, w1, w2;
  int simple_weight, oneref_noscale;
  int weight_x[MAX_BLK], weight_y[MAX_BLK];
  int16_t obmc[MAX_BLK][MAX_BLK];
  uint8_t block[MAX_BLK][MAX_BLK];
  uint8_t bref[2][MAX_BLK][MAX_BLK];
  int16_t *acc;
  int acc_stride;
} Motion;

#define ACC(m,x,y) ((int16_t *)((char *)(m)->acc + (ptrdiff_t)(m)->acc_stride*(y)) + (x))

/* __schro_upsampled_frame_get_subdata_prec1, schroframe.c:2187-2201 */
static const uint8_t *
subdata_prec1 (const OracleUpComp * c, int x, int y)
{
  int i = ((y & 1) << 1) | (x & 1);
  x >>= 1;
  y >>= 1;
  return c->plane[i] + (ptrdiff_t) c->stride * y + x;
}

/* schro_upsampled_frame_get_block_fast_prec3, schroframe.c:2288-2413 */
static void
get_block_prec3 (const Motion * m, const OracleUpComp * c, int x, int y,
    uint8_t dst[MAX_BLK][MAX_BLK])
{
  int hx = x >> 2, hy = y >> 2;
  int rx = x & 3, ry = y & 3;
  int ii, jj;
  int st = c->stride;

  switch ((ry << 2) | rx) {
    case 0:{
      const uint8_t *s = subdata_prec1 (c, hx, hy);
      for (jj = 0; jj < m->yblen; jj++)
        for (ii = 0; ii < m->xblen; ii++)
          dst[jj][ii] = s[(ptrdiff_t) st * jj + ii];
      break;
    }
    case 2:
    case 8:{
      const uint8_t *a = subdata_prec1 (c, hx, hy);
      const uint8_t *b = (rx == 0) ? subdata_prec1 (c, hx, hy + 1)
          : subdata_prec1 (c, hx + 1, hy);
      /* orc_avg2_*_u8: avgub */
      for (jj = 0; jj < m->yblen; jj++)
        for (ii = 0; ii < m->xblen; ii++)
          dst[jj][ii] = (uint8_t) ((a[(ptrdiff_t) st * jj + ii] +
                  b[(ptrdiff_t) st * jj + ii] + 1) >> 1);
      break;
    }
    default:{
      int w00 = (4 - ry) * (4 - rx);
      int w01 = (4 - ry) * rx;
      int w10 = ry * (4 - rx);
      int w11 = ry * rx;
      const uint8_t *p00 = subdata_prec1 (c, hx, hy);
      const uint8_t *p01 = subdata_prec1 (c, hx + 1, hy);
      const uint8_t *p10 = subdata_prec1 (c, hx, hy + 1);
      const uint8_t *p11 = subdata_prec1 (c, hx + 1, hy + 1);
      /* orc_combine4_nxm_u8, schroorc.orc:1635-1662: 16-bit mullw/addw */
      for (jj = 0; jj < m->yblen; jj++)
        for (ii = 0; ii < m->xblen; ii++) {
          ptrdiff_t o = (ptrdiff_t) st * jj + ii;
          int16_t t2 = (int16_t) (p00[o] * w00);
          t2 = (int16_t) (t2 + (int16_t) (p01[o] * w01));
          t2 = (int16_t) (t2 + (int16_t) (p10[o] * w10));
          t2 = (int16_t) (t2 + (int16_t) (p11[o] * w11));
          t2 = (int16_t) (t2 + 8);
          t2 = (int16_t) (t2 >> 4);
          dst[jj][ii] = (uint8_t) clampi (t2, 0, 255);
        }
      break;
    }
  }
}

/* get_block, schromotion8.c:303-335 + get_block_fast_precN,
 * schroframe.c:2459-2482 */
static void
get_block (Motion * m, int ref, int i, int j, int dx, int dy)
{
  const OracleUpComp *c = m->ref[ref];
  int x, y, px, py, exp;
  int ii, jj;

  if (m->k > 0) {
    dx >>= m->p->chroma_h_shift;
    dy >>= m->p->chroma_v_shift;
  }
  x = m->xbsep * i - m->xoffset;
  y = m->ybsep * j - m->yoffset;
  px = x * (1 << m->prec) + dx;
  py = y * (1 << m->prec) + dy;
  exp = 32 << m->prec;
  px = clampi (px, -exp, m->max_fast_x + exp - 1);
  py = clampi (py, -exp, m->max_fast_y + exp - 1);

  switch (m->prec) {
    case 0:{
      const uint8_t *s = c->plane[0] + (ptrdiff_t) c->stride * py + px;
      for (jj = 0; jj < m->yblen; jj++)
        for (ii = 0; ii < m->xblen; ii++)
          m->bref[ref][jj][ii] = s[(ptrdiff_t) c->stride * jj + ii];
      break;
    }
    case 1:{
      const uint8_t *s = subdata_prec1 (c, px, py);
      for (jj = 0; jj < m->yblen; jj++)
        for (ii = 0; ii < m->xblen; ii++)
          m->bref[ref][jj][ii] = s[(ptrdiff_t) c->stride * jj + ii];
      break;
    }
    case 2:
      get_block_prec3 (m, c, px * 2, py * 2, m->bref[ref]);
      break;
    default:
      get_block_prec3 (m, c, px, py, m->bref[ref]);
      break;
  }
}

static const OracleMotionVector *
mv_at (const Motion * m, int i, int j)
{
  return &m->mvs[j * m->p->x_num_blocks + i];
}

/* schro_motion_block_predict_block, schromotion8.c:542-568 (edge blocks) */
static void
predict_block (Motion * m, int i, int j)
{
  const OracleMotionVector *mv = mv_at (m, i, j);
  int mode = mv->flags & 3;
  int ii, jj;

  switch (mode) {
    case 0:                    /* get_dc_block :337-355 */
      for (jj = 0; jj < m->yblen; jj++)
        for (ii = 0; ii < m->xblen; ii++)
          m->block[jj][ii] = (uint8_t) (mv->v[m->k] + 128);
      break;
    case 1:
    case 2:{                   /* get_ref1_block / get_ref2_block :369-440 */
      int r = mode - 1;
      int weight = m->w1 + m->w2;
      get_block (m, r, i, j, mv->v[r], mv->v[2 + r]);
      for (jj = 0; jj < m->yblen; jj++)
        for (ii = 0; ii < m->xblen; ii++) {
          int s = m->bref[r][jj][ii];
          if (m->oneref_noscale)
            m->block[jj][ii] = (uint8_t) s;
          else                  /* ROUND_SHIFT stored into a uint8_t */
            m->block[jj][ii] =
                (uint8_t) ((s * weight + (1 << (m->bits - 1))) >> m->bits);
        }
      break;
    }
    default:                   /* get_biref_block :457-539 */
      get_block (m, 0, i, j, mv->v[0], mv->v[2]);
      get_block (m, 1, i, j, mv->v[1], mv->v[3]);
      for (jj = 0; jj < m->yblen; jj++)
        for (ii = 0; ii < m->xblen; ii++) {
          int a = m->bref[0][jj][ii], b = m->bref[1][jj][ii];
          if (m->simple_weight) {
            m->block[jj][ii] = (uint8_t) ((a + b + 1) >> 1);    /* avgub */
          } else {
            /* orc_combine2_nxm_u8 (w1, w2, (1<<bits)>>1, bits),
             * schroorc.orc:1737-1757 */
            int16_t t1 = (int16_t) (a * m->w1);
            int16_t t2 = (int16_t) (b * m->w2);
            t1 = (int16_t) (t1 + t2);
            t1 = (int16_t) (t1 + (int16_t) ((1 << m->bits) >> 1));
            t1 = (int16_t) (t1 >> m->bits);
            m->block[jj][ii] = (uint8_t) clampi (t1, 0, 255);
          }
        }
      break;
  }
}

/* schro_motion_block_accumulate_slow, schromotion8.c:659-698 */
static void
accumulate_slow (Motion * m, int x, int y)
{
  int i, j, w_x, w_y;
  for (j = 0; j < m->yblen; j++) {
    if (y + j < 0 || y + j >= m->height)
      continue;
    w_y = m->weight_y[j];
    if (y + j < m->yoffset)
      w_y += m->weight_y[2 * m->yoffset - j - 1];
    if (y + j >= m->p->y_num_blocks * m->ybsep - m->yoffset)
      w_y += m->weight_y[2 * (m->yblen - m->yoffset) - j - 1];
    for (i = 0; i < m->xblen; i++) {
      int16_t *d;
      if (x + i < 0 || x + i >= m->width)
        continue;
      w_x = m->weight_x[i];
      if (x + i < m->xoffset)
        w_x += m->weight_x[2 * m->xoffset - i - 1];
      if (x + i >= m->p->x_num_blocks * m->xbsep - m->xoffset)
        w_x += m->weight_x[2 * (m->xblen - m->xoffset) - i - 1];
      d = ACC (m, x + i, y + j);
      *d = (int16_t) (*d + m->block[j][i] * w_x * w_y);
    }
  }
}

/* schro_motion_block_predict_and_acc, schromotion8.c:570-657 (interior) */
static void
predict_and_acc (Motion * m, int x, int y, int i, int j)
{
  const OracleMotionVector *mv = mv_at (m, i, j);
  int mode = mv->flags & 3;
  int ii, jj;

  if (mode == 1 || mode == 3)
    get_block (m, 0, i, j, mv->v[0], mv->v[2]);
  if (mode == 2 || mode == 3)
    get_block (m, 1, i, j, mv->v[1], mv->v[3]);

  for (jj = 0; jj < m->yblen; jj++) {
    for (ii = 0; ii < m->xblen; ii++) {
      int16_t *d = ACC (m, x + ii, y + jj);
      int16_t w = m->obmc[jj][ii];
      int16_t t1;
      if (mode == 0) {
        /* block_acc_dc: mullw s1, p1 (16-bit parameter) */
        t1 = (int16_t) (w * (int16_t) (mv->v[m->k] + 128));
      } else if (m->simple_weight) {
        if (mode == 3)          /* block_acc_avg: avgub, convubw, mullw */
          t1 = (int16_t) (((m->bref[0][jj][ii] + m->bref[1][jj][ii] +
                      1) >> 1) * w);
        else                    /* block_acc: convubw, mullw */
          t1 = (int16_t) (m->bref[mode - 1][jj][ii] * w);
      } else if (mode == 3) {
        /* block_acc_biref, p1 = w1 << (6-bits), p2 = w2 << (6-bits) */
        int16_t p1 = (int16_t) (m->w1 << (6 - m->bits));
        int16_t p2 = (int16_t) (m->w2 << (6 - m->bits));
        int16_t t2;
        t1 = (int16_t) (m->bref[0][jj][ii] * p1);
        t2 = (int16_t) (m->bref[1][jj][ii] * p2);
        t1 = (int16_t) (t1 + t2);
        t1 = (int16_t) (t1 + 32);
        t1 = (int16_t) (t1 >> 6);
        t1 = (int16_t) (t1 * w);
      } else {
        /* block_acc_scaled, p1 = (w1+w2) << (6-bits) */
        int16_t p1 = (int16_t) ((m->w1 + m->w2) << (6 - m->bits));
        t1 = (int16_t) (m->bref[mode - 1][jj][ii] * p1);
        t1 = (int16_t) (t1 + 32);
        t1 = (int16_t) (t1 >> 6);
        t1 = (int16_t) (t1 * w);
      }
      *d = (int16_t) (*d + t1);
    }
  }
}

static void
zero_rows (Motion * m, int y, int n)
{
  int j;
  for (j = 0; j < n; j++)
    memset (ACC (m, 0, y + j), 0, sizeof (int16_t) * (size_t) m->width);
}

/* orc_rrshift6_add_s16_2d / _s32_2d, schroorc.orc:636-661 */
static void
finalize_rows (Motion * m, uint8_t * out, int out_stride,
    const void *residual, int res_stride, int res_bpp, int y, int n)
{
  int j, x;
  for (j = 0; j < n; j++) {
    uint8_t *o = out + (ptrdiff_t) out_stride * (y + j);
    const char *r = (const char *) residual + (ptrdiff_t) res_stride * (y + j);
    const int16_t *a = ACC (m, 0, y + j);
    for (x = 0; x < m->width; x++) {
      int16_t t1 = (int16_t) (a[x] + 32);
      int16_t s1;
      t1 = (int16_t) (t1 >> 6);
      if (res_bpp == 2)
        s1 = ((const int16_t *) r)[x];
      else
        s1 = (int16_t) ((const int32_t *) r)[x];        /* convlw */
      t1 = (int16_t) (s1 + t1);
      o[x] = (uint8_t) clampi (t1, 0, 255);
    }
  }
}

int
oracle_motion_render_u8 (const OracleMotionVector * mvs,
    const OracleMotionParams * p, int k,
    const OracleUpComp * ref1, const OracleUpComp * ref2,
    const void *residual, int res_stride, int res_bpp,
    int16_t * acc, int acc_stride,
    uint8_t * out, int out_stride, int width, int height)
{
  Motion *m = (Motion *) calloc (1, sizeof (Motion));
  int i, j, x, y, max_x_blocks, max_y_blocks;

  m->mvs = mvs;
  m->p = p;
  m->k = k;
  m->ref[0] = ref1;
  m->ref[1] = ref2;
  m->acc = acc;
  m->acc_stride = acc_stride;
  m->width = width;
  m->height = height;
  m->prec = p->mv_precision;
  m->bits = p->picture_weight_bits;
  m->w1 = p->picture_weight_1;
  m->w2 = p->picture_weight_2;

  m->xbsep = p->xbsep_luma;
  m->ybsep = p->ybsep_luma;
  m->xblen = p->xblen_luma;
  m->yblen = p->yblen_luma;
  if (k > 0) {
    m->xbsep >>= p->chroma_h_shift;
    m->ybsep >>= p->chroma_v_shift;
    m->xblen >>= p->chroma_h_shift;
    m->yblen >>= p->chroma_v_shift;
  }
  if (m->xblen > MAX_BLK || m->yblen > MAX_BLK || m->xblen < 1
      || m->yblen < 1 || m->prec < 0 || m->prec > 3) {
    free (m);
    return -1;
  }
  m->xoffset = (m->xblen - m->xbsep) / 2;
  m->yoffset = (m->yblen - m->ybsep) / 2;
  m->max_fast_x = (width - m->xblen) * (1 << m->prec);
  m->max_fast_y = (height - m->yblen) * (1 << m->prec);
  m->simple_weight = (m->w1 == 1 && m->w2 == 1 && m->bits == 1);
  m->oneref_noscale = (m->w1 + m->w2 == (1 << m->bits));

  init_weights (m->weight_x, m->xblen, m->xoffset);
  init_weights (m->weight_y, m->yblen, m->yoffset);
  for (j = 0; j < m->yblen; j++)
    for (i = 0; i < m->xblen; i++)
      m->obmc[j][i] = (int16_t) (m->weight_x[i] * m->weight_y[j]);

  max_x_blocks = p->x_num_blocks - 1;
  if ((width - m->xoffset) / m->xbsep < max_x_blocks)
    max_x_blocks = (width - m->xoffset) / m->xbsep;
  max_y_blocks = p->y_num_blocks - 1;
  if ((height - m->yoffset) / m->ybsep < max_y_blocks)
    max_y_blocks = (height - m->yoffset) / m->ybsep;

  /* block row 0, :793-827 */
  j = 0;
  zero_rows (m, 0, m->ybsep + m->yoffset);
  for (i = 0; i < p->x_num_blocks; i++) {
    x = m->xbsep * i - m->xoffset;
    y = m->ybsep * j - m->yoffset;
    predict_block (m, i, j);
    accumulate_slow (m, x, y);
  }
  finalize_rows (m, out, out_stride, residual, res_stride, res_bpp, 0,
      m->ybsep - m->yoffset);

  /* interior block rows, :828-872 */
  for (j = 1; j < max_y_blocks; j++) {
    y = m->ybsep * j - m->yoffset;
    zero_rows (m, y + m->yoffset * 2, m->ybsep);
    i = 0;
    x = m->xbsep * i - m->xoffset;
    predict_block (m, i, j);
    accumulate_slow (m, x, y);
    for (i = 1; i < max_x_blocks; i++) {
      x = m->xbsep * i - m->xoffset;
      predict_and_acc (m, x, y, i, j);
    }
    for (; i < p->x_num_blocks; i++) {
      x = m->xbsep * i - m->xoffset;
      predict_block (m, i, j);
      accumulate_slow (m, x, y);
    }
    finalize_rows (m, out, out_stride, residual, res_stride, res_bpp, y,
        m->ybsep);
  }

  /* bottom block rows, :873-903 */
  for (j = max_y_blocks; j < p->y_num_blocks; j++) {
    y = m->ybsep * j - m->yoffset;
    zero_rows (m, y + m->yoffset * 2,
        clampi (height - (y + m->yoffset * 2), 0, m->ybsep));
    for (i = 0; i < p->x_num_blocks; i++) {
      x = m->xbsep * i - m->xoffset;
      predict_block (m, i, j);
      accumulate_slow (m, x, y);
    }
    finalize_rows (m, out, out_stride, residual, res_stride, res_bpp, y,
        clampi (height - y, 0, m->ybsep));
  }

  /* last partial rows, :905-922 */
  y = p->y_num_blocks * m->ybsep - m->yoffset;
  finalize_rows (m, out, out_stride, residual, res_stride, res_bpp, y,
      clampi (height - y, 0, m->ybsep));

  free (m);
  return 0;
}
