/*
 * oracle_wavelet_tmpl.h -- included twice by oracle_wavelet.c (T = int16_t,
 * int32_t).  TEST INFRASTRUCTURE; see schro_oracle.h.
 *
 * Width of every intermediate follows the Orc opcode lists in
 * schroedinger/schroorc.orc (C bodies: schroorc-dist.c under DISABLE_ORC):
 *   s16: pair sums wrap to 16 bits BEFORE the widening multiply (addw then
 *        mulswl/convswl, :232-237, :308-316); products and rounding in 32 bits;
 *        the result is truncated (convlw) and added with 16-bit wrap.
 *   s32: everything wraps at 32 bits (addl, mulll, subl, :1815-2164).
 *   avgsw/avgsl never wrap.
 */
#define PASTE2(a,b) a##_##b
#define PASTE(a,b) PASTE2(a,b)
#define FN(name) PASTE(name, SUF)

/* value of one lifting term from its neighbours s[0..ntaps) */
static inline T
FN (lift_term) (const Step * st, const T * s)
{
  switch (st->kind) {
    case K_ADD2_22:{
      T t = WRAP (WADD32 (s[0], s[1]));
      t = WRAP (WADD32 (t, 2));
      return (T) (t >> 2);
    }
    case K_AVG11:
      return (T) AVG (s[0], s[1]);
    case K_MAS4:{
      T t1 = WRAP (WADD32 (s[1], s[2]));
      int32_t t3 = WMUL (t1, 9);
      T t2 = WRAP (WADD32 (s[0], s[3]));
      t3 = (int32_t) ((uint32_t) t3 - (uint32_t) (int32_t) t2);
      t3 = WADD32 (t3, st->rnd);
      t3 >>= st->sh;
      return WRAP (t3);
    }
    case K_HAAR_HALF:
      return (T) AVG (s[0], 0);
    case K_HAAR_FULL:
      return s[0];
    case K_MAS8:{
      int32_t x = st->rnd;
      int k;
      for (k = 0; k < 8; k++)
        x = WADD32 (x, WMUL (s[k], st->taps[k]));
      return WRAP (x >> 8);
    }
    case K_MAS2:{
      T t1 = WRAP (WADD32 (s[0], s[1]));
      int32_t t2 = WMUL (t1, st->c);
      t2 = WADD32 (t2, st->rnd);
      t2 >>= st->sh;
      return WRAP (t2);
    }
  }
  return 0;
}

/* d[x] +/-= term(src rows) for a whole row; dir = +1 inverse, -1 forward.
 * One plain loop per step kind so that the compiler can vectorise it (this file
 * is also the CPU baseline bench.py prints next to the GPU number). */
#define ROW_LOOP(EXPR)                                                        \
  do {                                                                        \
    if (sign > 0)                                                             \
      for (x = 0; x < n; x++) { T t = (EXPR); d[x] = WRAP (WADD32 (d[x], t)); } \
    else                                                                      \
      for (x = 0; x < n; x++) { T t = (EXPR);                                 \
        d[x] = WRAP ((int32_t) ((uint32_t) (int32_t) d[x] - (uint32_t) (int32_t) t)); } \
  } while (0)

static void
FN (lift_row) (const Step * st, int dir, T * ORACLE_RESTRICT d, const T * const *src, int n)
{
  int sign = st->sign * dir;
  int x;
  const T *s0 = src[0], *s1 = src[1], *s2 = src[2], *s3 = src[3];
  switch (st->kind) {
    case K_ADD2_22:
      ROW_LOOP ((T) (WRAP (WADD32 (WRAP (WADD32 (s0[x], s1[x])), 2)) >> 2));
      break;
    case K_AVG11:
      ROW_LOOP ((T) AVG (s0[x], s1[x]));
      break;
    case K_MAS4:{
      const int rnd = st->rnd, sh = st->sh;
      ROW_LOOP (WRAP (WADD32 ((int32_t) ((uint32_t) WMUL (WRAP (WADD32 (s1[x], s2[x])), 9)
                      - (uint32_t) (int32_t) WRAP (WADD32 (s0[x], s3[x]))), rnd) >> sh));
      break;
    }
    case K_HAAR_HALF:
      ROW_LOOP ((T) AVG (s0[x], 0));
      break;
    case K_HAAR_FULL:
      ROW_LOOP (s0[x]);
      break;
    case K_MAS2:{
      const int c = st->c, rnd = st->rnd, sh = st->sh;
      ROW_LOOP (WRAP (WADD32 (WMUL (WRAP (WADD32 (s0[x], s1[x])), c), rnd) >> sh));
      break;
    }
    default:{                  /* K_MAS8 */
      T s[8];
      int k;
      for (x = 0; x < n; x++) {
        T t;
        for (k = 0; k < 8; k++)
          s[k] = src[k][x];
        t = FN (lift_term) (st, s);
        if (sign > 0)
          d[x] = WRAP (WADD32 (d[x], t));
        else
          d[x] = WRAP ((int32_t) ((uint32_t) (int32_t) d[x] - (uint32_t) (int32_t) t));
      }
      break;
    }
  }
}

#undef ROW_LOOP

/* one lifting step over two 1-D arrays A[n], B[n] (horizontal direction): the
 * other array is copied with 8 replicated samples on both sides (what extend_N_M
 * does, schrowaveletorc.c:192-269) and the row kernel above runs on shifted views */
static void
FN (lift_1d) (const Step * st, int dir, T * A, T * B, int n, T * pad)
{
  int nt = ntaps (st->kind);
  T *d = st->target ? B : A;
  const T *o = st->target ? A : B;
  const T *src[8];
  int k;
  T *e = pad + 8;
  memcpy (e, o, sizeof (T) * (size_t) n);
  for (k = 1; k <= 8; k++) {
    e[-k] = o[0];
    e[n - 1 + k] = o[n - 1];
  }
  for (k = 0; k < 8; k++)
    src[k] = e + st->off + (k < nt ? k : 0);
  FN (lift_row) (st, dir, d, src, n);
}

/* one lifting step vertically: A = even rows, B = odd rows of the view */
static void
FN (lift_vert) (const Step * st, int dir, T * data, int stride, int width,
    int height)
{
  int n = height / 2;
  int nt = ntaps (st->kind);
  int r, k;
  const T *src[8];
  for (r = 0; r < n; r++) {
    T *d = (T *) ((char *) data + (size_t) stride * (2 * r + st->target));
    for (k = 0; k < 8; k++) {
      int rr = clampi (r + st->off + (k < nt ? k : 0), 0, n - 1);
      src[k] = (const T *) ((const char *) data +
          (size_t) stride * (2 * rr + (1 - st->target)));
    }
    FN (lift_row) (st, dir, d, src, width);
  }
}

static void
FN (iiwt_2d) (T * data, int stride, int width, int height, const Filter * f)
{
  int k, y, i;
  int n = width / 2;
  T *A = (T *) malloc (sizeof (T) * ((size_t) width + n + 16));
  T *B = A + n;
  T *pad = A + width;

  for (k = 0; k < f->nsteps; k++)
    FN (lift_vert) (&f->steps[k], +1, data, stride, width, height);

  for (y = 0; y < height; y++) {
    T *line = (T *) ((char *) data + (size_t) stride * y);
    memcpy (A, line, sizeof (T) * (size_t) n);
    memcpy (B, line + n, sizeof (T) * (size_t) n);
    for (k = 0; k < f->nsteps; k++)
      FN (lift_1d) (&f->steps[k], +1, A, B, n, pad);
    /* orc_interleave2_rrshift1_* (schroorc.orc:770-781,1866-1877): add wraps;
     * orc_haar_synth_rrshift1_int_* (:1000-1013): avgs*, no wrap;
     * orc_interleave2_* / orc_haar_synth_int_*: no shift. */
    for (i = 0; i < n; i++) {
      T a = A[i], b = B[i];
      if (f->shift == 1) {
        a = (T) (WRAP (WADD32 (a, 1)) >> 1);
        b = (T) (WRAP (WADD32 (b, 1)) >> 1);
      } else if (f->shift == 2) {
        a = (T) AVG (a, 0);
        b = (T) AVG (b, 0);
      }
      line[2 * i] = a;
      line[2 * i + 1] = b;
    }
  }
  free (A);
}

static void
FN (iwt_2d) (T * data, int stride, int width, int height, const Filter * f)
{
  int k, y, i;
  int n = width / 2;
  T *A = (T *) malloc (sizeof (T) * ((size_t) width + n + 16));
  T *B = A + n;
  T *pad = A + width;

  /* horizontal first (wavelet_iwt_*_horiz, e.g. schrowaveletorc.c:288-301):
   * orc_deinterleave2_lshift1_* for the filters with an output shift */
  for (y = 0; y < height; y++) {
    T *line = (T *) ((char *) data + (size_t) stride * y);
    for (i = 0; i < n; i++) {
      T a = line[2 * i], b = line[2 * i + 1];
      if (f->shift) {
        a = WRAP (WMUL (a, 2));
        b = WRAP (WMUL (b, 2));
      }
      A[i] = a;
      B[i] = b;
    }
    for (k = f->nsteps - 1; k >= 0; k--)
      FN (lift_1d) (&f->steps[k], -1, A, B, n, pad);
    memcpy (line, A, sizeof (T) * (size_t) n);
    memcpy (line + n, B, sizeof (T) * (size_t) n);
  }
  for (k = f->nsteps - 1; k >= 0; k--)
    FN (lift_vert) (&f->steps[k], -1, data, stride, width, height);
  free (A);
}

#undef PASTE2
#undef PASTE
#undef FN
