/*
 * oracle_frame.c -- CPU restatement of the reference-frame side of the path:
 * MC edge extension, 8-tap half-pel upsampling, s16/s32 -> u8 convert.
 * TEST INFRASTRUCTURE (see schro_oracle.h); never linked into the product.
 *
 * Follows schroedinger/schroframe.c: mas8_u8_edgeextend :1515-1555,
 * schro_frame_upsample_horiz :1557-1574, schro_frame_upsample_vert
 * :1612-1645, schro_frame_mc_edgeextend_{horiz,vert} :1940-1985,
 * schro_upsampled_frame_upsample :2000-2030.  The aprons are materialised
 * exactly as the reference does (incl. its cross-plane apron sources), so the
 * GPU path's "clamp the half-pel coordinate instead" is something the parity
 * tests prove rather than assume.
 */
#include "schro_oracle.h"
#include <stdlib.h>
#include <stddef.h>
#include <string.h>

static const int up_taps[8] = { -1, 3, -7, 21, 21, -7, 3, -1 };

static inline int
clampi (int x, int lo, int hi)
{
  return x < lo ? lo : (x > hi ? hi : x);
}

OracleUpComp *
oracle_upcomp_new (int width, int height, int ext)
{
  OracleUpComp *c = (OracleUpComp *) calloc (1, sizeof (OracleUpComp));
  size_t plane_bytes;
  int i;
  c->width = width;
  c->height = height;
  c->ext = ext;
  c->stride = (width + 2 * ext + 15) & ~15;
  plane_bytes = (size_t) c->stride * (size_t) (height + 2 * ext);
  c->alloc = (uint8_t *) calloc (4, plane_bytes);
  for (i = 0; i < 4; i++)
    c->plane[i] = c->alloc + plane_bytes * i + (size_t) c->stride * ext + ext;
  return c;
}

void
oracle_upcomp_free (OracleUpComp * c)
{
  if (!c)
    return;
  free (c->alloc);
  free (c);
}

void
oracle_upcomp_set_plane0 (OracleUpComp * c, const uint8_t * src,
    int src_stride)
{
  int y;
  for (y = 0; y < c->height; y++)
    memcpy (c->plane[0] + (size_t) c->stride * y,
        src + (size_t) src_stride * y, (size_t) c->width);
}

void
oracle_upcomp_get_plane (const OracleUpComp * c, int i, uint8_t * dst,
    int dst_stride)
{
  int y;
  for (y = 0; y < c->height; y++)
    memcpy (dst + (size_t) dst_stride * y,
        c->plane[i] + (ptrdiff_t) c->stride * y, (size_t) c->width);
}

int
oracle_upcomp_get (const OracleUpComp * c, int i, int x, int y)
{
  return c->plane[i][(ptrdiff_t) c->stride * y + x];
}

/* schro_frame_mc_edgeextend_horiz, schroframe.c:1940-1957 */
static void
edgeextend_horiz (const OracleUpComp * c, uint8_t * frame,
    const uint8_t * src)
{
  int j;
  int width = c->width, ext = c->ext;
  for (j = 0; j < c->height; j++) {
    uint8_t *line = frame + (ptrdiff_t) c->stride * j;
    const uint8_t *src_line = src + (ptrdiff_t) c->stride * j;
    uint8_t left = src_line[0], right = src_line[width - 1];
    memset (line - ext, left, (size_t) ext);
    /* "Remember to overwrite the last horizontal pel" */
    memset (line + width - 1, right, (size_t) ext + 1);
  }
}

/* schro_frame_mc_edgeextend_vert, schroframe.c:1959-1985 */
static void
edgeextend_vert (const OracleUpComp * c, uint8_t * frame, const uint8_t * src)
{
  int j;
  int width = c->width, height = c->height, ext = c->ext;
  size_t len = (size_t) width + 2 * (size_t) ext;
  for (j = 0; j < ext; j++) {
    memmove (frame + (ptrdiff_t) c->stride * (-j - 1) - ext, src - ext, len);
    memmove (frame + (ptrdiff_t) c->stride * (height + j) - ext,
        src + (ptrdiff_t) c->stride * (height - 1) - ext, len);
  }
  /* "Copy the src into the bottom line of frame" */
  memmove (frame + (ptrdiff_t) c->stride * (height - 1) - ext,
      src + (ptrdiff_t) c->stride * (height - 1) - ext, len);
}

void
oracle_upcomp_edgeextend (OracleUpComp * c)
{
  edgeextend_horiz (c, c->plane[0], c->plane[0]);
  edgeextend_vert (c, c->plane[0], c->plane[0]);
}

/* mas8_u8_edgeextend (..., taps, 16, 5, 3, n), schroframe.c:1515-1555 */
static void
upsample_row (uint8_t * d, const uint8_t * s, int n, int apron)
{
  int i, j;
  if (apron >= 4 && n > 8) {
    /* the row is already edge-extended (replicated), so s[-3..n+3] IS the clamped
     * read; plain loop the compiler can vectorise (CPU baseline of bench.py) */
    for (i = 0; i < n - 1; i++) {
      int x = 21 * (s[i] + s[i + 1]) - 7 * (s[i - 1] + s[i + 2])
          + 3 * (s[i - 2] + s[i + 3]) - (s[i - 3] + s[i + 4]);
      x = (x + 16) >> 5;
      d[i] = (uint8_t) (x < 0 ? 0 : (x > 255 ? 255 : x));
    }
    d[n - 1] = s[n - 1];
    return;
  }
  for (i = 0; i < n; i++) {
    int x = 0;
    for (j = 0; j < 8; j++)
      x += s[clampi (i + j - 3, 0, n - 1)] * up_taps[j];
    d[i] = (uint8_t) clampi ((x + 16) >> 5, 0, 255);
  }
  if (n > 8)
    d[n - 1] = s[n - 1];        /* only the n > 8 branch copies the last pel */
}

/* schro_frame_upsample_horiz, schroframe.c:1557-1574 */
static void
upsample_horiz (const OracleUpComp * c, uint8_t * dest, const uint8_t * src)
{
  int j;
  for (j = 0; j < c->height; j++)
    upsample_row (dest + (ptrdiff_t) c->stride * j,
        src + (ptrdiff_t) c->stride * j, c->width, c->ext);
}

/* schro_frame_upsample_vert, schroframe.c:1612-1645 */
static void
upsample_vert (const OracleUpComp * c, uint8_t * dest, const uint8_t * src)
{
  int i, j, k;
  int height = c->height, width = c->width;
  for (j = 0; j < height - 1; j++) {
    uint8_t *d = dest + (ptrdiff_t) c->stride * j;
    const uint8_t *r[8];
    for (k = 0; k < 8; k++)
      r[k] = src + (ptrdiff_t) c->stride * clampi (j + k - 3, 0, height - 1);
    for (i = 0; i < width; i++) {
      int x = 21 * (r[3][i] + r[4][i]) - 7 * (r[2][i] + r[5][i])
          + 3 * (r[1][i] + r[6][i]) - (r[0][i] + r[7][i]);
      x = (x + 16) >> 5;
      d[i] = (uint8_t) (x < 0 ? 0 : (x > 255 ? 255 : x));
    }
  }
  j = height - 1;
  memcpy (dest + (ptrdiff_t) c->stride * j, src + (ptrdiff_t) c->stride * j,
      (size_t) width);
}

/* schro_upsampled_frame_upsample, schroframe.c:2000-2030 (one component) */
void
oracle_upcomp_upsample (OracleUpComp * c)
{
  uint8_t **fd = c->plane;

  upsample_vert (c, fd[2], fd[0]);
  edgeextend_horiz (c, fd[2], fd[2]);
  edgeextend_vert (c, fd[2], fd[0]);

  upsample_horiz (c, fd[1], fd[0]);
  edgeextend_horiz (c, fd[1], fd[0]);
  edgeextend_vert (c, fd[1], fd[1]);

  upsample_horiz (c, fd[3], fd[2]);
  edgeextend_horiz (c, fd[3], fd[2]);
  edgeextend_vert (c, fd[3], fd[1]);
}

/* convert_u8_s16 / convert_u8_s32 + crop, schrovirtframe.c:1689-1720,1853;
 * orc_offsetconvert_u8_s16: addw 128 (wraps) then convsuswb;
 * orc_offsetconvert_u8_s32: addl 128, convssslw (saturate to s16), convsuswb
 * (schroorc.orc:504-521). */
void
oracle_convert_u8_from_signed (uint8_t * dst, int dst_stride,
    const void *src, int src_stride, int bpp, int width, int height)
{
  int x, y;
  for (y = 0; y < height; y++) {
    uint8_t *d = dst + (size_t) dst_stride * y;
    const char *sl = (const char *) src + (size_t) src_stride * y;
    for (x = 0; x < width; x++) {
      int v;
      if (bpp == 2) {
        v = (int16_t) (((const int16_t *) sl)[x] + 128);
      } else {
        int32_t t =
            (int32_t) ((uint32_t) ((const int32_t *) sl)[x] + 128u);
        v = clampi (t, -32768, 32767);
      }
      d[x] = (uint8_t) clampi (v, 0, 255);
    }
  }
}
