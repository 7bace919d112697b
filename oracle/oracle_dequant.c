/* oracle_dequant.c -- CPU restatement (TEST INFRASTRUCTURE, never shipped or linked by the
 * product) of the core-syntax coefficient reconstruction AFTER entropy decoding:
 *
 *   schrodecoder.c:3311-3322  schro_decoder_zero_block (orc_splat_s16_2d / _s32_2d with 0)
 *   schrodecoder.c:3400-3451  schro_decoder_decode_codeblock_noarith: quantiser tables, then
 *                             orc_dequantise_s16_2d_8xn / _4xn / _s16_ip_2d (16-bit Orc
 *                             arithmetic) for s16 frames, orc_dequantise_s32_ip_2d for s32
 *   schrodecoder.c:3060-3083  the arithmetic-coded line decoders' inline dequantisation
 *                             v = (quant_offset + quant_factor * v + 2) >> 2, sign, store
 *   schrodecoder.c:3482-3487  quant_factor = schro_table_quant[i], quant_offset =
 *                             schro_table_offset_3_8[i] (inter) / _1_2[i] (intra)
 *
 * Pinned by: the reference's compiled orc_dequantise_s16_2d_8xn / _4xn / _s16_ip_2d /
 * _s32_ip_2d (oracle/_ref, tests/test_oracle_dequant.py); the three tables against the
 * reference's numbers (tests/golden/quant_tables.json, arith_lut.json); and, as a whole, by
 * the reference's own stream: the quantised values of test_stream.drc go through this code
 * and must give the coefficients whose decoded pictures carry the reference decoder's MD5s. */
#include <stdint.h>
#include <string.h>
#include "schro_oracle.h"

/* schro_table_offset_3_8 (schrotables.c): (3 * factor + 4) / 8, entry 0 is 1 */
int
oracle_quant_offset_3_8 (int q)
{
  if (q == 0)
    return 1;
  return (int) ((oracle_quant_factor (q) * 3 + 4) / 8);
}

static int32_t
load_q (const void *src, int src_bytes, size_t n)
{
  switch (src_bytes) {
    case 1: return ((const int8_t *) src)[n];
    case 2: return ((const int16_t *) src)[n];
    default: return ((const int32_t *) src)[n];
  }
}

/* One codeblock.  src: its quantised values, row-major and tight (width * src_bytes per row),
 * or NULL for a zero codeblock.  arith: 0 = C int arithmetic of the arithmetic-coded decoders
 * and of orc_dequantise_s32_*, 1 = the 16-bit Orc arithmetic of the VLC (noarith) s16 path. */
void
oracle_dequant_codeblock (void *dst, int dst_stride, int bpp, const void *src, int src_bytes, int width,
    int height, int quant_index, int is_intra, int arith)
{
  const int qi = quant_index < 0 ? 0 : (quant_index > 60 ? 60 : quant_index);
  const int factor = (int) oracle_quant_factor (qi);
  const int offset = is_intra ? (int) oracle_quant_offset_1_2 (qi) : oracle_quant_offset_3_8 (qi);
  for (int y = 0; y < height; y++) {
    char *row = (char *) dst + (size_t) y * dst_stride;
    for (int x = 0; x < width; x++) {
      int32_t v = 0;
      if (src) {
        const int32_t q = load_q (src, src_bytes, (size_t) y * width + x);
        if (arith == 1) {
          /* copyw, signw, absw, mullw p1, addw p2, shrsw 2, mullw sign (schroorc.orc:1098-1170);
           * p1 = quant_factor, p2 = quant_offset + 2, both used as 16-bit parameters */
          const int16_t qs = (int16_t) q;
          const int16_t sign = qs > 0 ? 1 : (qs < 0 ? -1 : 0);
          const int16_t mag = (int16_t) (qs < 0 ? -qs : qs);
          int16_t t = (int16_t) (mag * (int16_t) factor);
          t = (int16_t) (t + (int16_t) (offset + 2));
          t = (int16_t) (t >> 2);
          v = (int16_t) (t * sign);
        } else if (q) {
          const uint32_t mag = q < 0 ? 0u - (uint32_t) q : (uint32_t) q;
          const int32_t d = (int32_t) (mag * (uint32_t) factor + (uint32_t) offset + 2u) >> 2;
          v = q < 0 ? (int32_t) (0u - (uint32_t) d) : d;
        }
      }
      if (bpp == 2)
        ((int16_t *) row)[x] = (int16_t) v;
      else
        ((int32_t *) row)[x] = v;
    }
  }
}
