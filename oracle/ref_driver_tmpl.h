/* ref_driver_tmpl.h -- included twice by ref_driver.c.  TEST INFRASTRUCTURE. */
#define PASTE2(a,b) a##_##b
#define PASTE(a,b) PASTE2(a,b)
#define FN(name) PASTE(name, SUF)

static void
FN (extend) (T * a, int n, int nl, int nr)
{
  int k;
  for (k = 1; k <= nl; k++)
    a[-k] = a[0];
  for (k = 0; k < nr; k++)
    a[n + k] = a[n - 1];
}

static void
FN (refdrv_horiz) (T * line, int width, int filter, T * tmp)
{
  int n = width / 2;
  T *hi = tmp + 8;              /* even / low-pass half  */
  T *lo = tmp + n + 24;         /* odd / high-pass half */

  memcpy (hi, line, sizeof (T) * (size_t) n);
  memcpy (lo, line + n, sizeof (T) * (size_t) n);
  switch (filter) {
    case 0:
      FN (extend) (lo, n, 2, 2);
      H_ADD2_SUB (hi, lo - 1, n);
      FN (extend) (hi, n, 2, 2);
      H_MAS4_ADD (lo, hi - 1, 1 << 3, 4, n);
      ILV_RR (line, hi, lo, n);
      break;
    case 1:
      FN (extend) (lo, n, 1, 1);
      H_ADD2_SUB (hi, lo - 1, n);
      FN (extend) (hi, n, 1, 1);
      H_AVG_ADD (lo, hi, n);
      ILV_RR (line, hi, lo, n);
      break;
    case 2:
      FN (extend) (lo, n, 2, 1);
      H_MAS4_SUB (hi, lo - 2, 1 << 4, 5, n);
      FN (extend) (hi, n, 1, 2);
      H_MAS4_ADD (lo, hi - 1, 1 << 3, 4, n);
      ILV_RR (line, hi, lo, n);
      break;
    case 3:
      HAAR_INT (line, hi, lo, n);
      break;
    case 4:
      HAAR_INT_RR (line, hi, lo, n);
      break;
    case 6:
      FN (extend) (lo, n, 1, 1);
      MAS2_SUB_IP (hi, lo - 1, 1817, 2048, 12, n);
      FN (extend) (hi, n, 1, 1);
      MAS2_SUB_IP (lo, hi, 3616, 2048, 12, n);
      FN (extend) (lo, n, 1, 1);
      MAS2_ADD_IP (hi, lo - 1, 217, 2048, 12, n);
      FN (extend) (hi, n, 1, 1);
      MAS2_ADD_IP (lo, hi, 6497, 2048, 12, n);
      ILV_RR (line, hi, lo, n);
      break;
  }
}

static void
FN (refdrv_iiwt) (T * data, int stride, int width, int height, int filter)
{
#define ROW(y) ((T *)((char *)data + (size_t)stride * (size_t)(y)))
  T *tmp = (T *) calloc ((size_t) width + 64, sizeof (T));
  int i, j;
  int h = height;
  int ahead1, ahead2;

  if (filter == 6) {
    for (j = 0; j < h; j += 2)
      MAS2_SUB_OP (ROW (j), ROW (j), ROW (j == 0 ? 1 : j - 1), ROW (j + 1),
          1817, 2048, 12, width);
    for (j = 1; j < h; j += 2)
      MAS2_SUB_OP (ROW (j), ROW (j), ROW (j - 1),
          ROW (j + 1 < h ? j + 1 : j - 1), 3616, 2048, 12, width);
    for (j = 0; j < h; j += 2)
      MAS2_ADD_OP (ROW (j), ROW (j), ROW (j == 0 ? 1 : j - 1), ROW (j + 1),
          217, 2048, 12, width);
    for (j = 1; j < h; j += 2)
      MAS2_ADD_OP (ROW (j), ROW (j), ROW (j - 1),
          ROW (j + 1 < h ? j + 1 : j - 1), 6497, 2048, 12, width);
    for (j = 0; j < h; j++)
      FN (refdrv_horiz) (ROW (j), width, filter, tmp);
    free (tmp);
    return;
  }

  switch (filter) {
    case 0:
      ahead1 = 7;
      ahead2 = 3;
      break;
    case 1:
      ahead1 = 2;
      ahead2 = 1;
      break;
    case 2:
      ahead1 = 8;
      ahead2 = 4;
      break;
    default:
      ahead1 = 1;
      ahead2 = 0;
      break;
  }

  for (i = -ahead1; i < h; i++) {
    /* first vertical lifting step, `ahead1` rows in front of the output row */
    j = i + ahead1;
    if (j >= 0 && j < h && !(j & 1)) {
      switch (filter) {
        case 0:
        case 1:
          V_ADD2_SUB (ROW (j), ROW (j == 0 ? 1 : j - 1), ROW (j + 1), width);
          break;
        case 2:
          V_MAS4_SUB (ROW (j), ROW (clampi (j - 3, 1, h - 1)),
              ROW (clampi (j - 1, 1, h - 1)), ROW (clampi (j + 1, 1, h - 1)),
              ROW (clampi (j + 3, 1, h - 1)), 1 << 4, 5, width);
          break;
        default:
          HAAR_V (ROW (j), ROW (j + 1), width);
          break;
      }
    }
    /* second vertical lifting step */
    j = i + ahead2;
    if (filter <= 2 && j >= 0 && j < h && (j & 1)) {
      if (filter == 1) {
        V_AVG_ADD (ROW (j), ROW (j - 1), ROW (j + 1 < h ? j + 1 : j - 1),
            width);
      } else {
        V_MAS4_ADD (ROW (j), ROW (clampi (j - 3, 0, h - 2)),
            ROW (clampi (j - 1, 0, h - 2)), ROW (clampi (j + 1, 0, h - 2)),
            ROW (clampi (j + 3, 0, h - 2)), 1 << 3, 4, width);
      }
    }
    /* horizontal synthesis + interleave of the finished row */
    j = i;
    if (j >= 0 && j < h)
      FN (refdrv_horiz) (ROW (j), width, filter, tmp);
  }
  free (tmp);
#undef ROW
}

#undef PASTE2
#undef PASTE
#undef FN
