/*
 * oracle_lowdelay.c -- CPU restatement of the VC-2 low-delay transform-data decode:
 * per-slice exp-Golomb unpack + dequantisation into the interleaved coefficient frame,
 * then DC prediction of the LL band.  TEST INFRASTRUCTURE (see schro_oracle.h).
 *
 * Follows, in dschleef/schroedinger 1.0.11.1:
 *   schrolowdelay.c:109-187   schro_decoder_decode_slice_slow       (s16, ragged slices)
 *   schrolowdelay.c:189-268   schro_decoder_decode_slice_slow_s32   (s32)
 *   schrolowdelay.c:271-310   schro_decoder_decode_slice_fast       (s16, even slices)
 *   schrolowdelay.c:559-762   the three slice loops and the choice between them
 *   schrodecoder.c:3219-3277  schro_decoder_subband_dc_predict (_s32)
 *   schrounpack.c:16-271      the bit reader (restated by bit position, see below)
 *   schroutils.c:180-189      schro_dequantise
 *   schroparams.c:319-352     schro_subband_get_frame_data, :355-368 sub-band order
 *   schroframe.c:1865-1884    schro_frame_data_get_codeblock
 *   schrotables.c             schro_table_quant / schro_table_offset_1_2 (generated here by
 *                             the Dirac spec formula; tests pin all 61 entries of both
 *                             against the reference's constants, tests/golden/)
 *
 * PARITY STATUS: "parity unpinned" for the slice syntax as a whole -- the reference has no
 * golden slice data and schrolowdelay.c / schrounpack.c do not compile here (they include
 * schro.h -> orc/orc.h; no stand-ins).  Pinned pieces: the 16-bit dequantisation of the
 * fast path against the compiled orc_dequantise_var_s16_ip (oracle/_ref), the tables
 * against the reference's constants, the exp-Golomb reader against the reader that parses the
 * reference's test stream (oracle/dirac_stream.py, validated by that stream's digests).
 *
 * The reference's SchroUnpack is a 32-bit shift register over a byte pointer with a count
 * of bits left; what a caller observes is "bit number pos of the buffer, or the guard bit
 * (1 for slices) once pos reaches the end".  schro_unpack_limit_bits_remaining moves the
 * end to pos + n -- past the slice if a corrupt slice_y_length says so, the reference then
 * reads the following slice's bytes; this restatement stops at the end of the whole buffer
 * (where the reference would read out of bounds).  schro_unpack_decode_sint_s16's table
 * walk returns the same values as the bit-by-bit form for every code it can hold.
 * 1 << count with count >= 31 is undefined in the reference; here (and on the GPU) the
 * arithmetic is modulo 2^32 with 1 << count taken as 0 for count >= 32.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "schro_oracle.h"

typedef struct {
  const uint8_t *data;
  int64_t pos, end;             /* bit positions */
} Bits;

static inline unsigned
get_bit (Bits * b)
{
  unsigned v = 1;               /* guard bit, schrolowdelay.c:126 */
  if (b->pos < b->end)
    v = (b->data[b->pos >> 3] >> (7 - (b->pos & 7))) & 1;
  b->pos++;
  return v;
}

static unsigned
get_bits (Bits * b, int n)
{
  unsigned v = 0;
  while (n-- > 0)
    v = (v << 1) | get_bit (b);
  return v;
}

/* schro_unpack_decode_uint / _sint_slow, schrounpack.c:211-241 */
static int32_t
get_sint (Bits * b)
{
  uint32_t count = 0, value = 0;
  while (!get_bit (b)) {
    count++;
    value = (value << 1) | get_bit (b);
  }
  value += (count < 32 ? (uint32_t) 1 << count : 0) - 1;
  if (value && get_bit (b))
    value = 0u - value;
  return (int32_t) value;
}

uint32_t
oracle_quant_factor (int q)
{
  /* Dirac specification 13.3.1 quant_factor (); the reference stores the results */
  uint64_t base = (uint64_t) 1 << (q / 4);
  switch (q & 3) {
    case 0:
      return (uint32_t) (4 * base);
    case 1:
      return (uint32_t) ((503829 * base + 52958) / 105917);
    case 2:
      return (uint32_t) ((665857 * base + 58854) / 117708);
    default:
      return (uint32_t) ((440253 * base + 32722) / 65444);
  }
}

uint32_t
oracle_quant_offset_1_2 (int q)
{
  if (q == 0)
    return 1;
  if (q == 1)
    return 2;
  return (oracle_quant_factor (q) + 1) / 2;
}

/* schroutils.c:180-189 */
static inline int
dequantise (int q, int quant_factor, int quant_offset)
{
  if (q == 0)
    return 0;
  /* int arithmetic; a product beyond 31 bits (no legal stream) wraps as gcc's does */
  if (q < 0)
    return -((int32_t) ((uint32_t) - q * (uint32_t) quant_factor + (uint32_t) quant_offset + 2u) >> 2);
  return (int32_t) ((uint32_t) q * (uint32_t) quant_factor + (uint32_t) quant_offset + 2u) >> 2;
}

/* orc_dequantise_var_s16_ip, schroorc.orc:1204-1217: every step in 16 bits.  The factor
 * and offset + 2 are stored as int16_t (schrolowdelay.c:478-479) */
int16_t
oracle_dequantise_var_s16 (int16_t q, int quant_factor, int quant_offset)
{
  const int16_t f = (int16_t) quant_factor, o = (int16_t) (quant_offset + 2);
  const int16_t sign = q > 0 ? 1 : (q < 0 ? -1 : 0);
  const int16_t mag = (int16_t) (q < 0 ? -q : q);       /* absw: -32768 stays -32768 */
  int16_t t = (int16_t) (mag * f);
  t = (int16_t) (t + o);
  t = (int16_t) (t >> 2);
  return (int16_t) (t * sign);
}

typedef struct {
  uint8_t *data;
  int stride, width, height;
} Band;

/* schro_subband_get_frame_data, schroparams.c:319-352 */
static Band
subband (void *data, int stride, int iwt_w, int iwt_h, int depth, int index, int bpp)
{
  static const int subband_position[] = { 0, 1, 2, 3, 5, 6, 7, 9, 10, 11, 13, 14, 15,
    17, 18, 19, 21, 22, 23, 25, 26, 27
  };
  const int position = subband_position[index];
  const int shift = depth - (position >> 2);
  Band b;
  b.stride = stride << shift;
  b.width = iwt_w >> shift;
  b.height = iwt_h >> shift;
  b.data = (uint8_t *) data;
  if (position & 2)
    b.data += b.stride >> 1;
  if (position & 1)
    b.data += b.width * bpp;
  return b;
}

/* schro_frame_data_get_codeblock, schroframe.c:1865-1884 */
static Band
codeblock (const Band * src, int x, int y, int nh, int nv, int bpp)
{
  const int xmin = (src->width * x) / nh, xmax = (src->width * (x + 1)) / nh;
  const int ymin = (src->height * y) / nv, ymax = (src->height * (y + 1)) / nv;
  Band b;
  b.data = src->data + (size_t) ymin * src->stride + (size_t) xmin * bpp;
  b.stride = src->stride;
  b.width = xmax - xmin;
  b.height = ymax - ymin;
  return b;
}

static int
ilog2up (unsigned int x)
{                               /* schrolowdelay.c:94-105 */
  int i;
  for (i = 0; i < 32; i++) {
    if (x == 0)
      return i;
    x >>= 1;
  }
  return 0;
}

static inline void
put (uint8_t * line, int x, int bpp, int v)
{
  if (bpp == 2)
    ((int16_t *) line)[x] = (int16_t) v;
  else
    ((int32_t *) line)[x] = v;
}

int
oracle_lowdelay_arith (const OracleLowDelayParams * p, int bpp)
{
  /* schro_decoder_decode_lowdelay_transform_data, schrolowdelay.c:746-762 */
  if (bpp == 4)
    return ORACLE_LOWDELAY_S32;
  if ((p->iwt_chroma_width >> p->transform_depth) % p->n_horiz_slices == 0 &&
      (p->iwt_chroma_height >> p->transform_depth) % p->n_vert_slices == 0)
    return ORACLE_LOWDELAY_FAST16;
  return ORACLE_LOWDELAY_SLOW16;
}

int
oracle_lowdelay_decode (const uint8_t * data, int64_t n_data_bytes, void *const comp[3],
    const int stride[3], const OracleLowDelayParams * p, int bpp)
{
  const int nsub = 1 + 3 * p->transform_depth;
  const int arith = oracle_lowdelay_arith (p, bpp);
  const int n_bytes = p->slice_bytes_num / p->slice_bytes_denom;
  const int remainder = p->slice_bytes_num % p->slice_bytes_denom;
  int accumulator = 0;
  int64_t offset = 0;
  int sx, sy, i, k, x, y;

  if (bpp != 2 && bpp != 4)
    return -1;
  if (p->transform_depth < 0 || p->transform_depth > 6 || p->n_horiz_slices < 1 || p->n_vert_slices < 1
      || p->slice_bytes_denom < 1)
    return -1;

  for (sy = 0; sy < p->n_vert_slices; sy++) {
    for (sx = 0; sx < p->n_horiz_slices; sx++) {
      int extra = 0, slice_bytes, base_index, length_bits, slice_y_length;
      Bits yb, uvb;
      accumulator += remainder; /* schrolowdelay.c:615-623 */
      if (accumulator >= p->slice_bytes_denom) {
        extra = 1;
        accumulator -= p->slice_bytes_denom;
      }
      slice_bytes = n_bytes + extra;
      if (offset + slice_bytes > n_data_bytes)
        return -2;

      yb.data = data;
      yb.pos = 8 * offset;
      yb.end = 8 * (offset + slice_bytes);
      base_index = (int) get_bits (&yb, 7);
      /* the fast path sizes the length field once, from the short slice (:577) */
      length_bits = ilog2up (8u * (unsigned) (arith == ORACLE_LOWDELAY_FAST16 ? n_bytes : slice_bytes));
      slice_y_length = (int) get_bits (&yb, length_bits);
      uvb = yb;
      yb.end = yb.pos + slice_y_length;         /* schro_unpack_limit_bits_remaining */
      if (yb.end > 8 * n_data_bytes)
        yb.end = 8 * n_data_bytes;
      uvb.pos += slice_y_length;                /* schro_unpack_skip_bits */

      for (k = 0; k < 2; k++) {                 /* luma, then U and V interleaved */
        Bits *b = k ? &uvb : &yb;
        for (i = 0; i < nsub; i++) {
          const int qi = base_index - p->quant_matrix[i];
          const int quant_index = qi < 0 ? 0 : (qi > 60 ? 60 : qi);
          const int qf = (int) oracle_quant_factor (quant_index);
          const int qo = (int) oracle_quant_offset_1_2 (quant_index);
          Band sb[2], cb[2];
          int c;
          for (c = 0; c <= k; c++) {
            sb[c] = subband (comp[k + c], stride[k + c], k ? p->iwt_chroma_width : p->iwt_luma_width,
                k ? p->iwt_chroma_height : p->iwt_luma_height, p->transform_depth, i, bpp);
            cb[c] = codeblock (&sb[c], sx, sy, p->n_horiz_slices, p->n_vert_slices, bpp);
          }
          for (y = 0; y < cb[0].height; y++) {
            for (x = 0; x < cb[0].width; x++) {
              for (c = 0; c <= k; c++) {
                const int value = get_sint (b);
                const int v = arith == ORACLE_LOWDELAY_FAST16 ?
                    oracle_dequantise_var_s16 ((int16_t) value, qf, qo) : dequantise (value, qf, qo);
                put (cb[c].data + (size_t) y * cb[c].stride, x, bpp, v);
              }
            }
          }
        }
      }
      offset += slice_bytes;
    }
  }

  for (k = 0; k < 3; k++) {
    Band ll = subband (comp[k], stride[k], k ? p->iwt_chroma_width : p->iwt_luma_width,
        k ? p->iwt_chroma_height : p->iwt_luma_height, p->transform_depth, 0, bpp);
    oracle_dc_predict (ll.data, ll.stride, ll.width, ll.height, bpp);
  }
  return 0;
}

/* schrodecoder.c:3219-3277 */
void
oracle_dc_predict (void *data, int stride, int width, int height, int bpp)
{
  int i, j;
  if (bpp == 2) {
    for (j = 0; j < height; j++) {
      int16_t *line = (int16_t *) ((uint8_t *) data + (size_t) j * stride);
      int16_t *prev = (int16_t *) ((uint8_t *) data + (size_t) (j - 1) * stride);
      if (j == 0) {
        for (i = 1; i < width; i++)
          line[i] = (int16_t) (line[i] + line[i - 1]);
        continue;
      }
      line[0] = (int16_t) (line[0] + prev[0]);
      for (i = 1; i < width; i++) {
        const int a = line[i - 1] + prev[i] + prev[i - 1] + 1;
        line[i] = (int16_t) (line[i] + ((a * 21845 + 10922) >> 16));    /* schro_divide3, schroutils.h:64 */
      }
    }
  } else {
    for (j = 0; j < height; j++) {
      int32_t *line = (int32_t *) ((uint8_t *) data + (size_t) j * stride);
      int32_t *prev = (int32_t *) ((uint8_t *) data + (size_t) (j - 1) * stride);
      if (j == 0) {
        for (i = 1; i < width; i++)
          line[i] = (int32_t) ((uint32_t) line[i] + (uint32_t) line[i - 1]);
        continue;
      }
      line[0] = (int32_t) ((uint32_t) line[0] + (uint32_t) prev[0]);
      for (i = 1; i < width; i++) {
        const int32_t a = (int32_t) ((uint32_t) line[i - 1] + (uint32_t) prev[i] + (uint32_t) prev[i - 1] + 1u);
        const int32_t d = a < 0 ? (a - 3 + 1) / 3 : a / 3;      /* schro_divide (a, 3), schroutils.h:63 */
        line[i] = (int32_t) ((uint32_t) line[i] + (uint32_t) d);
      }
    }
  }
}

/* ---- test-vector generator: the inverse of oracle_lowdelay_decode's slice syntax --------
 * (not a restatement of the reference's encoder, schrolowdelay.c:764-; a plain writer of
 * the syntax the decoder above reads).  comp[] hold QUANTISED values as int32_t in the
 * coefficient frame layout (bpp is the sample size of the picture they stand for: it
 * selects the decoder and with it the width of the slice_y_length field); every slice gets base_index[slice], its luma codes, slice_y_length = the
 * luma bits actually written, then the chroma codes.  Codes that do not fit the slice are
 * cut off at its last bit (the decoder then reads guard bits); unused bits are pad_bit.
 * y_length_bias is added to every slice_y_length field (0 for a legal stream). */
typedef struct {
  uint8_t *data;
  int64_t pos, end;
} BitSink;

static void
put_bit (BitSink * b, unsigned v)
{
  if (b->pos < b->end) {
    const int sh = 7 - (int) (b->pos & 7);
    b->data[b->pos >> 3] = (uint8_t) ((b->data[b->pos >> 3] & ~(1u << sh)) | ((v & 1u) << sh));
  }
  b->pos++;
}

static void
put_sint (BitSink * b, int32_t v)
{
  const uint64_t m = (uint64_t) (v < 0 ? 0u - (uint32_t) v : (uint32_t) v) + 1u;        /* |v| + 1 */
  int count = 0, k;
  while ((m >> (count + 1)) != 0)
    count++;
  for (k = count - 1; k >= 0; k--) {
    put_bit (b, 0);
    put_bit (b, (unsigned) ((m >> k) & 1u));
  }
  put_bit (b, 1);
  if (v)
    put_bit (b, v < 0);
}

int
oracle_lowdelay_write (uint8_t * data, int64_t n_data_bytes, void *const comp[3], const int stride[3],
    const OracleLowDelayParams * p, int bpp, const uint8_t * base_index, int pad_bit, int y_length_bias)
{
  const int nsub = 1 + 3 * p->transform_depth;
  const int arith = oracle_lowdelay_arith (p, bpp);
  const int n_bytes = p->slice_bytes_num / p->slice_bytes_denom;
  const int remainder = p->slice_bytes_num % p->slice_bytes_denom;
  int accumulator = 0;
  int64_t offset = 0;
  int sx, sy, i, k, x, y, c;

  for (sy = 0; sy < p->n_vert_slices; sy++) {
    for (sx = 0; sx < p->n_horiz_slices; sx++) {
      int extra = 0, slice_bytes, length_bits;
      int64_t field_pos, y_start, y_bits;
      BitSink b;
      accumulator += remainder;
      if (accumulator >= p->slice_bytes_denom) {
        extra = 1;
        accumulator -= p->slice_bytes_denom;
      }
      slice_bytes = n_bytes + extra;
      if (offset + slice_bytes > n_data_bytes)
        return -2;
      memset (data + offset, pad_bit ? 0xff : 0, (size_t) slice_bytes);
      b.data = data;
      b.pos = 8 * offset;
      b.end = 8 * (offset + slice_bytes);
      for (k = 6; k >= 0; k--)
        put_bit (&b, (base_index[sy * p->n_horiz_slices + sx] >> k) & 1u);
      length_bits = ilog2up (8u * (unsigned) (arith == ORACLE_LOWDELAY_FAST16 ? n_bytes : slice_bytes));
      field_pos = b.pos;
      b.pos += length_bits;
      y_start = b.pos;
      for (k = 0; k < 2; k++) {
        for (i = 0; i < nsub; i++) {
          Band sb[2], cb[2];
          for (c = 0; c <= k; c++) {
            sb[c] = subband (comp[k + c], stride[k + c], k ? p->iwt_chroma_width : p->iwt_luma_width,
                k ? p->iwt_chroma_height : p->iwt_luma_height, p->transform_depth, i, 4);
            cb[c] = codeblock (&sb[c], sx, sy, p->n_horiz_slices, p->n_vert_slices, 4);
          }
          for (y = 0; y < cb[0].height; y++)
            for (x = 0; x < cb[0].width; x++)
              for (c = 0; c <= k; c++)
                put_sint (&b, ((const int32_t *) (cb[c].data + (size_t) y * cb[c].stride))[x]);
        }
        if (k == 0) {
          /* luma bits that made it into the slice */
          BitSink f = b;
          y_bits = (b.pos < b.end ? b.pos : b.end) - y_start;
          y_bits += y_length_bias;
          if (y_bits < 0)
            y_bits = 0;
          f.pos = field_pos;
          for (c = length_bits - 1; c >= 0; c--)
            put_bit (&f, (unsigned) ((y_bits >> c) & 1));
        }
      }
      offset += slice_bytes;
    }
  }
  return 0;
}
