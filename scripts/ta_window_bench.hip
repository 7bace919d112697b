// microbenchmark: 24-byte x 24-row window gathers from a linear vs a 16x8-tiled image
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define G __attribute__ ((address_space (1)))
typedef uint32_t u32x3 __attribute__ ((ext_vector_type (3)));
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
typedef u32x3 u32x3_u __attribute__ ((aligned (1)));

// MODE 0: linear, 2 lanes x 12 B per row (48 lanes per window)
// MODE 1: tiled 16x8, aligned 16-B chunks, 3 chunks per row (72 lanes per window)
// MODE 2: tiled 16x8, 2 chunks per row only (x & 15 <= 8 case: 48 lanes)
// MODE 3: tiled 32x4 (32 B x 4 rows), 16-B chunks, 3 per row
template < int MODE >
__global__ __launch_bounds__ (256) void k (const uint8_t * img, const int2 * win, int nwin, int W, int iters, uint32_t * out)
{
  constexpr int LPW = MODE == 0 ? 48 : MODE == 2 ? 48 : 72;
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int l = tid % LPW;
  int w = tid / LPW;
  const int wstep = gridDim.x * 256 / LPW;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    const int2 o = win[w % nwin];
    if (MODE == 0) {
      const int row = l >> 1, half = l & 1;
      u32x3 v = *(const G u32x3_u *) (img + (size_t) (o.y + row) * W + o.x + 12 * half);
      acc += v.x ^ v.y ^ v.z;
    } else {
      constexpr int CPR = MODE == 2 ? 2 : 3;
      const int row = l / CPR, ch = l % CPR;
      const int x = (o.x & ~15) + 16 * ch, y = o.y + row;
      size_t a;
      if (MODE == 3) a = ((size_t) (y >> 2) * (W >> 5) + (x >> 5)) * 128 + (y & 3) * 32 + (x & 31);
      else a = ((size_t) (y >> 3) * (W >> 4) + (x >> 4)) * 128 + (y & 7) * 16;
      u32x4 v = *(const G u32x4 *) (img + a);
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    w += wstep;
  }
  out[tid] = acc;
}

int main (int argc, char **argv)
{
  const bool shuffle = argc > 1;
  const int W = 7680, H = 4320;         // one 2160p half-pel luma image (33 MB)
  uint8_t *img; (void) hipMalloc (&img, (size_t) W * H + 65536);
  (void) hipMemset (img, 1, (size_t) W * H + 65536);
  const int NW = 1 << 18;
  std::vector < int2 > win (NW);
  uint32_t s = 12345;
  // windows: blocks on an 8x8-pixel grid (16x16 half-pel), visited in raster order within
  // 128x32-pixel tiles, each displaced by a random vector of +-32 half-pel samples
  int n = 0;
  for (int ty = 0; ty < 40 && n < NW; ty++)
    for (int tx = 0; tx < 28 && n < NW; tx++)
      for (int by = 0; by < 4; by++)
        for (int bx = 0; bx < 16 && n < NW; bx++) {
          s = s * 1664525u + 1013904223u; int dx = (int) ((s >> 8) % 65) - 32;
          s = s * 1664525u + 1013904223u; int dy = (int) ((s >> 8) % 65) - 32;
          int x = 64 + tx * 256 + bx * 16 + dx, y = 64 + ty * 64 + by * 16 + dy;
          win[n++] = make_int2 (x, y);
        }
  if (shuffle) {                // emulate the mode-sorted block order: random within each 64-window tile
    for (int t0 = 0; t0 + 64 <= n; t0 += 64)
      for (int i = 63; i > 0; i--) {
        s = s * 1664525u + 1013904223u;
        int j = (s >> 8) % (i + 1);
        int2 tmp = win[t0 + i]; win[t0 + i] = win[t0 + j]; win[t0 + j] = tmp;
      }
  }
  int2 *d_win; (void) hipMalloc (&d_win, n * 8); (void) hipMemcpy (d_win, win.data (), n * 8, hipMemcpyHostToDevice);
  uint32_t *out; (void) hipMalloc (&out, 4 * 256 * 16384);
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  const char *names[] = { "linear, x3 2 lanes/row", "tiled 16x8, 3 chunks/row", "tiled 16x8, 2 chunks/row", "tiled 32x4, 3 chunks/row" };
  printf ("%d windows%s\n", n, shuffle ? " (shuffled within tiles)" : "");
  for (int mode = 0; mode < 4; mode++) {
    const int grid = 9 * 1024, iters = 8;
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord (e0);
      switch (mode) {
        case 0: k < 0 ><<< grid, 256 >>> (img, d_win, n, W, iters, out); break;
        case 1: k < 1 ><<< grid, 256 >>> (img, d_win, n, W, iters, out); break;
        case 2: k < 2 ><<< grid, 256 >>> (img, d_win, n, W, iters, out); break;
        case 3: k < 3 ><<< grid, 256 >>> (img, d_win, n, W, iters, out); break;
      }
      hipEventRecord (e1); hipEventSynchronize (e1);
      hipEventElapsedTime (&ms, e0, e1);
    }
    const int LPW = mode == 0 || mode == 2 ? 48 : 72;
    double windows = (double) grid * 256 / LPW * iters;
    printf ("%-28s %8.3f ms  %7.3f Gwindows/s\n", names[mode], ms, windows / ms / 1e6);
  }
  return 0;
}
