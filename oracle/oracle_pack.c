/*
 * oracle_pack.c -- CPU restatement of the copy-out of a decoded u8 picture into a
 * packed output frame: schro_frame_convert (dest packed, src planar u8),
 * schroedinger/schroframe.c:869-979.  TEST INFRASTRUCTURE (see schro_oracle.h).
 *
 * The reference builds a chain of line-rendering virtual frames:
 *   unpack (identity for planar)                      schrovirtframe.c:902
 *   subsample to the packed format's chroma (4:2:2 for YUYV/UYVY, 4:4:4 for AYUV):
 *     nearest neighbour, no filtering                 convert_4xx_4yy :1438-1537
 *   crop (dest smaller) / edge-extend (dest larger): component line min(i, h-1),
 *     columns past the source replicate its last     crop_u8 :1823-1831, edge_extend_u8 :1881-1895
 *   pack                                              pack_yuyv :943-957 (orc_packyuyv,
 *     schroorc.orc:718-734), pack_uyvy :972-991, pack_ayuv :1230-1247
 * and renders it line by line.  Composed per output sample below.
 */
#include "schro_oracle.h"

static inline int
mini (int a, int b)
{
  return a < b ? a : b;
}

static inline int
round_up_shift (int x, int s)
{
  return (x + (1 << s) - 1) >> s;
}

/* component `comp` sample (X, Y) of the virtual frame just before packing */
static int
chain_sample (const OraclePackSrc * s, int comp, int t_hs, int X, int Y)
{
  int x, y;
  if (comp == 0) {
    x = mini (X, s->width - 1);         /* edge_extend_u8 / crop_u8 */
    y = mini (Y, s->height - 1);
  } else {
    /* size of the subsampled source's chroma component (schro_frame_new_virtual dims) */
    int sw = round_up_shift (s->width, t_hs), sh = s->height;   /* target v_shift is 0 */
    int Xc = mini (X, sw - 1), Yc = mini (Y, sh - 1);
    /* convert_420_422 / _420_444 / _422_444 / _444_422 */
    if (t_hs == s->h_shift)
      x = Xc;
    else if (t_hs > s->h_shift)
      x = 2 * Xc;
    else
      x = Xc >> 1;
    y = s->v_shift ? Yc >> 1 : Yc;
  }
  return s->data[comp][(long) y * s->stride[comp] + x];
}

/* ---- v210 --------------------------------------------------------------------
 * schro_frame_convert with a v210 destination (schroframe.c:889-891, 960-962):
 *   u8 source:  chain as above to U8_422, then pack_v210 (schrovirtframe.c:1129-1210),
 *               10-bit value = (x << 2) | (x >> 6);
 *   s16 / s32 source: must already be 4:2:2 (schro_virt_frame_new_subsample only knows
 *               the u8 formats, :1545-1575); s32 is first truncated to 16 bits
 *               (convert_s16_s32 -> orc_convert_s16_s32 = convlw, schroorc.orc:483-487);
 *               crop_s16 / edge_extend_s16 (:1833-1841, 1897-1912); pack_v210_s16
 *               (:1044-1127), 10-bit value = clamp (x + 512, 0, 1023).
 * Six pixels make four little-endian words; a row is ceil (width / 6) such groups, the
 * samples of the last group beyond `width` are 0. */
static int
v210_sample (const OraclePackSrc * s, int bpp, int comp, int X, int Y)
{
  int x, y;
  long v;
  if (bpp == 1)
    v = chain_sample (s, comp, 1, X, Y);
  else {
    int cw = comp ? round_up_shift (s->width, 1) : s->width;
    x = mini (X, cw - 1);
    y = mini (Y, s->height - 1);
    if (bpp == 2)
      v = ((const int16_t *) (s->data[comp] + (long) y * s->stride[comp]))[x];
    else
      v = (int16_t) ((const int32_t *) (s->data[comp] + (long) y * s->stride[comp]))[x];
  }
  if (bpp == 1)
    return (int) ((v << 2) | (v >> 6));
  v += 512;
  return (int) (v < 0 ? 0 : (v > 1023 ? 1023 : v));
}

int
oracle_pack_v210 (uint8_t * dst, int dst_stride, int width, int height,
    const OraclePackSrc * s, int src_bpp)
{
  int i, j, k;
  if (!dst || !s || width <= 0 || height <= 0 || s->width <= 0 || s->height <= 0)
    return -1;
  if (src_bpp != 1 && src_bpp != 2 && src_bpp != 4)
    return -1;
  if (src_bpp != 1 && !(s->h_shift == 1 && s->v_shift == 0))
    return -1;
  if ((width < s->width || height < s->height) && (width > s->width || height > s->height))
    return -1;
  for (i = 0; i < height; i++) {
    uint8_t *d = dst + (long) i * dst_stride;
    for (j = 0; j * 6 < width; j++) {
      uint32_t yv[6], cb[3], cr[3], w[4];
      for (k = 0; k < 6; k++)
        yv[k] = (j * 6 + k) < width ? (uint32_t) v210_sample (s, src_bpp, 0, j * 6 + k, i) : 0;
      for (k = 0; k < 3; k++) {
        cb[k] = (j * 6 + 2 * k) < width ? (uint32_t) v210_sample (s, src_bpp, 1, j * 3 + k, i) : 0;
        cr[k] = (j * 6 + 2 * k) < width ? (uint32_t) v210_sample (s, src_bpp, 2, j * 3 + k, i) : 0;
      }
      w[0] = (cr[0] << 20) | (yv[0] << 10) | cb[0];
      w[1] = (yv[2] << 20) | (cb[1] << 10) | yv[1];
      w[2] = (cb[2] << 20) | (yv[3] << 10) | cr[1];
      w[3] = (yv[5] << 20) | (cr[2] << 10) | yv[4];
      for (k = 0; k < 4; k++) {
        d[16 * j + 4 * k + 0] = (uint8_t) (w[k] & 0xff);
        d[16 * j + 4 * k + 1] = (uint8_t) ((w[k] >> 8) & 0xff);
        d[16 * j + 4 * k + 2] = (uint8_t) ((w[k] >> 16) & 0xff);
        d[16 * j + 4 * k + 3] = (uint8_t) ((w[k] >> 24) & 0xff);
      }
    }
  }
  return 0;
}

int
oracle_pack_u8 (uint8_t * dst, int dst_stride, int format, int width, int height,
    const OraclePackSrc * s)
{
  int i, j;
  if (!dst || !s || width <= 0 || height <= 0 || s->width <= 0 || s->height <= 0)
    return -1;
  /* the reference crops both dimensions or extends both (schroframe.c:931-941) */
  if ((width < s->width || height < s->height) && (width > s->width || height > s->height))
    return -1;
  for (i = 0; i < height; i++) {
    uint8_t *d = dst + (long) i * dst_stride;
    switch (format) {
      case ORACLE_FORMAT_YUYV:
      case ORACLE_FORMAT_UYVY:
        for (j = 0; j < width / 2; j++) {
          int y0 = chain_sample (s, 0, 1, 2 * j, i), y1 = chain_sample (s, 0, 1, 2 * j + 1, i);
          int u = chain_sample (s, 1, 1, j, i), v = chain_sample (s, 2, 1, j, i);
          if (format == ORACLE_FORMAT_YUYV) {
            d[4 * j + 0] = (uint8_t) y0;
            d[4 * j + 1] = (uint8_t) u;
            d[4 * j + 2] = (uint8_t) y1;
            d[4 * j + 3] = (uint8_t) v;
          } else {
            d[4 * j + 0] = (uint8_t) u;
            d[4 * j + 1] = (uint8_t) y0;
            d[4 * j + 2] = (uint8_t) v;
            d[4 * j + 3] = (uint8_t) y1;
          }
        }
        break;
      case ORACLE_FORMAT_AYUV:
        for (j = 0; j < width; j++) {
          d[4 * j + 0] = 0xff;
          d[4 * j + 1] = (uint8_t) chain_sample (s, 0, 0, j, i);
          d[4 * j + 2] = (uint8_t) chain_sample (s, 1, 0, j, i);
          d[4 * j + 3] = (uint8_t) chain_sample (s, 2, 0, j, i);
        }
        break;
      default:
        return -1;
    }
  }
  return 0;
}
