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

/* ---- v216, ARGB, AY64 ----------------------------------------------------------
 * schro_frame_convert with these destinations (schroframe.c:886-895): the source is first
 * brought to S16_422 (v216), S16_444 (ARGB) or S32_444 (AY64) --
 *   convert_s16_u8 / convert_s32_u8: x - 128 (orc_offsetconvert_s16_u8 / _s32_u8,
 *     schroorc.orc:524-540); convert_s16_s32: truncation (convlw :483-487);
 *     convert_s32_s16: sign extension (convswl :497-501)            schrovirtframe.c:1742-1817
 * -- its chroma format must already be the target's (schro_virt_frame_new_subsample knows the
 * u8 formats only and runs after the depth conversion, :1545-1575), then crop_s16 /
 * edge_extend_s16 and their s32 twins (:1833-1912: line min (i, h - 1), columns past the
 * source repeat its last sample), then
 *   pack_v216 (:1007-1028)   reads the S16 lines through uint8_t pointers: output bytes
 *                            8j+0,1 = BYTE j of the U line, 8j+2,3 = byte 2j of the Y line,
 *                            8j+4,5 = byte j of the V line, 8j+6,7 = byte 2j+1 of the Y line
 *                            (j < width / 2) -- the reference's behaviour, restated as it is;
 *   pack_argb (:1265-1287)   YCoCg-R: t = y + (cg >> 1), b = t - (co >> 1),
 *                            bytes 0xff, b + co, t + cg, b (stores truncate to 8 bits);
 *   pack_ayuv64 (:1302-1322) 16-bit words 0xffff, clamp (x + 0x8000, 0, 0xffff) for Y, U, V.
 * The reference holds no vectors for these and schrovirtframe.c cannot be compiled here
 * (orc.h): parity unpinned by reference output; the depth conversions are pinned on the
 * compiled orc_offsetconvert_s16_u8 / orc_convert_s16_s32 (tests/test_oracle_pack.py). */
static long
wide_sample (const OraclePackSrc * s, int bpp, int target_bpp, int t_hs, int comp, int X, int Y)
{
  const int cw = comp ? round_up_shift (s->width, t_hs) : s->width;
  const int x = mini (X, cw - 1), y = mini (Y, s->height - 1);
  const uint8_t *row = s->data[comp] + (long) y * s->stride[comp];
  long v;
  if (bpp == 1)
    v = (long) row[x] - 128;
  else if (bpp == 2)
    v = ((const int16_t *) row)[x];
  else
    v = ((const int32_t *) row)[x];
  return target_bpp == 2 ? (long) (int16_t) v : v;
}

int
oracle_pack_wide (uint8_t * dst, int dst_stride, int format, int width, int height,
    const OraclePackSrc * s, int src_bpp)
{
  int i, j;
  if (!dst || !s || width <= 0 || height <= 0 || s->width <= 0 || s->height <= 0)
    return -1;
  if (src_bpp != 1 && src_bpp != 2 && src_bpp != 4)
    return -1;
  if ((width < s->width || height < s->height) && (width > s->width || height > s->height))
    return -1;
  if (format == ORACLE_FORMAT_v216 ? !(s->h_shift == 1 && s->v_shift == 0) : !(s->h_shift == 0 && s->v_shift == 0))
    return -1;
  for (i = 0; i < height; i++) {
    uint8_t *d = dst + (long) i * dst_stride;
    switch (format) {
      case ORACLE_FORMAT_v216:
        for (j = 0; j < width / 2; j++) {
          /* byte b of an S16 line = half of sample b / 2 (little endian) */
#define LINE_BYTE(comp, b) ((uint8_t) ((uint16_t) wide_sample (s, src_bpp, 2, 1, comp, (b) >> 1, i) >> (8 * ((b) & 1))))
          const uint8_t u = LINE_BYTE (1, j), v = LINE_BYTE (2, j);
          const uint8_t y0 = LINE_BYTE (0, 2 * j), y1 = LINE_BYTE (0, 2 * j + 1);
#undef LINE_BYTE
          d[8 * j + 0] = d[8 * j + 1] = u;
          d[8 * j + 2] = d[8 * j + 3] = y0;
          d[8 * j + 4] = d[8 * j + 5] = v;
          d[8 * j + 6] = d[8 * j + 7] = y1;
        }
        break;
      case ORACLE_FORMAT_ARGB:
        for (j = 0; j < width; j++) {
          const int y = (int) wide_sample (s, src_bpp, 2, 0, 0, j, i);
          const int co = (int) wide_sample (s, src_bpp, 2, 0, 1, j, i);
          const int cg = (int) wide_sample (s, src_bpp, 2, 0, 2, j, i);
          const int t = y + (cg >> 1), b = t - (co >> 1);
          d[4 * j + 0] = 0xff;
          d[4 * j + 1] = (uint8_t) (b + co);
          d[4 * j + 2] = (uint8_t) (t + cg);
          d[4 * j + 3] = (uint8_t) b;
        }
        break;
      case ORACLE_FORMAT_AY64:
        for (j = 0; j < width; j++) {
          int k;
          uint16_t w[4];
          w[0] = 0xffff;
          for (k = 0; k < 3; k++) {
            long v = wide_sample (s, src_bpp, 4, 0, k, j, i) + 0x8000;
            w[1 + k] = (uint16_t) (v < 0 ? 0 : (v > 0xffff ? 0xffff : v));
          }
          for (k = 0; k < 4; k++) {
            d[8 * j + 2 * k] = (uint8_t) (w[k] & 0xff);
            d[8 * j + 2 * k + 1] = (uint8_t) (w[k] >> 8);
          }
        }
        break;
      default:
        return -1;
    }
  }
  return 0;
}

/* schro_frame_shift_right (schroframe.c:1265-1293) on one component, in place:
 * orc_add_const_rshift_s16 / _s32 (schroorc.orc:146-163): x = (x + ((1 << shift) >> 1)) >> shift,
 * the add wraps at the sample width, the shift is arithmetic.  The decoder applies it to an
 * intra picture's frame when the stream's bit depth exceeds the output picture's
 * (schrodecoder.c:2013-2019). */
int
oracle_shift_right (void *data, int stride, int width, int height, int bpp, int shift)
{
  int i, j;
  if (!data || width <= 0 || height <= 0 || (bpp != 2 && bpp != 4) || shift < 0 || shift >= 8 * bpp)
    return -1;
  for (j = 0; j < height; j++) {
    if (bpp == 2) {
      int16_t *p = (int16_t *) ((char *) data + (long) j * stride);
      for (i = 0; i < width; i++)
        p[i] = (int16_t) ((int16_t) (p[i] + (int16_t) ((1 << shift) >> 1)) >> shift);
    } else {
      int32_t *p = (int32_t *) ((char *) data + (long) j * stride);
      for (i = 0; i < width; i++)
        p[i] = (int32_t) ((uint32_t) p[i] + (uint32_t) ((1u << shift) >> 1)) >> shift;
    }
  }
  return 0;
}
