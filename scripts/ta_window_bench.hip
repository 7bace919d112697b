// Microbenchmark: the OBMC reference gather by itself.  1.5 M windows of 24 bytes x 24
// rows (one 12x12 block at quarter-pel on a 2x upsampled 2160p reference), each fetched
// once, in tile-major order with the blocks of a 128x32-pixel tile shuffled (the kernel
// sorts them by prediction mode), from a linear image and from a 16x8-byte tiled one.
//   hipcc --offload-arch=gfx950 -O3 scripts/ta_window_bench.hip -o build/ta_window_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define G __attribute__ ((address_space (1)))
typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
typedef u32x2 u32x2_u __attribute__ ((aligned (1)));

// MODE 0: linear, 8-byte unaligned loads, 3 lanes per row (the kernel's pattern): 72 lanes / window
// MODE 1: tiled 16x8, aligned 16-byte chunks, 3 chunks per row: 72 lanes / window
// MODE 2: tiled 16x8, chunks 0..1 always, chunk 2 only when the window needs it
// MODE 3: linear, but windows 8 rows apart collapsed: reads only every 8th row x 3 (line-count control)
template < int MODE >
__global__ __launch_bounds__ (256) void k (const uint8_t * img, const int2 * win, int nwin, int W, uint32_t * out)
{
  const int tid = blockIdx.x * 256 + threadIdx.x;
  constexpr int LPW = MODE == 4 ? 96 : 72;
  const int l = tid % LPW;
  const int w = tid / LPW;
  uint32_t acc = 0;
  if (w < nwin) {
    const int2 o = win[w];
    const int row = l / 3, ch = l % 3;
    if (MODE == 0) {
      u32x2 v = *(const G u32x2_u *) (img + (size_t) (o.y + row) * W + o.x + 8 * ch);
      acc = v.x ^ v.y;
    } else if (MODE == 3) {
      u32x2 v = *(const G u32x2_u *) (img + (size_t) (o.y + (row & ~7)) * W + o.x + 8 * ch);
      acc = v.x ^ v.y;
    } else if (MODE == 4) {     // 8 bytes x 16 rows per line, 4 aligned 8-byte chunks per row
      const int r4 = l >> 2, c4 = l & 3;
      const int x = (o.x & ~7) + 8 * c4, y = o.y + r4;
      const size_t a = ((size_t) (y >> 4) * (W >> 3) + (x >> 3)) * 128 + (y & 15) * 8;
      u32x2 v = *(const G u32x2 *) (img + a);
      acc = v.x ^ v.y;
    } else if (MODE == 5) {     // 32 bytes x 4 rows per line, 3 aligned 16-byte chunks per row
      const int x = (o.x & ~15) + 16 * ch, y = o.y + row;
      const size_t a = ((size_t) (y >> 2) * (W >> 5) + (x >> 5)) * 128 + (y & 3) * 32 + (x & 16);
      u32x4 v = *(const G u32x4 *) (img + a);
      acc = v.x ^ v.y ^ v.z ^ v.w;
    } else {
      const int x = (o.x & ~15) + 16 * ch, y = o.y + row;
      const bool need = MODE == 1 || ch < 2 || (o.x & 15) > 8;
      if (need) {
        const size_t a = ((size_t) (y >> 3) * (W >> 4) + (x >> 4)) * 128 + (y & 7) * 16;
        u32x4 v = *(const G u32x4 *) (img + a);
        acc = v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  if (acc == 0x12345678u)
    out[tid & 1023] = acc;
}

int main ()
{
  const int W = 7680, H = 4320;         // one 2160p half-pel luma image (33 MB); two of them
  uint8_t *img; (void) hipMalloc (&img, (size_t) 2 * W * H + 65536);
  (void) hipMemset (img, 1, (size_t) 2 * W * H + 65536);
  std::vector < int2 > win;
  uint32_t s = 12345;
  // 8 pictures x (30 x 68 tiles) x (16 x 4 blocks): block origin on the 16-sample grid
  // + a random vector of +-32 half-pel samples; alternating references
  for (int pic = 0; pic < 6; pic++)
    for (int ty = 0; ty < 66; ty++)
      for (int tx = 0; tx < 29; tx++) {
        const size_t t0 = win.size ();
        for (int by = 0; by < 4; by++)
          for (int bx = 0; bx < 16; bx++) {
            s = s * 1664525u + 1013904223u; int dx = (int) ((s >> 8) % 65) - 32;
            s = s * 1664525u + 1013904223u; int dy = (int) ((s >> 8) % 65) - 32;
            s = s * 1664525u + 1013904223u; int ref = (s >> 12) & 1;
            int x = 64 + tx * 256 + bx * 16 + dx, y = 40 + ty * 64 + by * 16 + dy + ref * H;
            win.push_back (make_int2 (x, y));
          }
        for (int i = 63; i > 0; i--) {
          s = s * 1664525u + 1013904223u;
          int j = (s >> 8) % (i + 1);
          int2 tmp = win[t0 + i]; win[t0 + i] = win[t0 + j]; win[t0 + j] = tmp;
        }
      }
  const int n = (int) win.size ();
  int2 *d_win; (void) hipMalloc (&d_win, (size_t) n * 8); (void) hipMemcpy (d_win, win.data (), (size_t) n * 8, hipMemcpyHostToDevice);
  uint32_t *out; (void) hipMalloc (&out, 4096);
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  const char *names[] = { "linear, 8 B x 3 lanes/row", "tiled 16x8, 3 chunks/row", "tiled 16x8, 2-3 chunks/row", "linear, every 8th row x3 (1/8 of the lines)", "tiled 8x16, 4 x 8 B chunks/row", "tiled 32x4, 3 chunks/row" };
  printf ("%d windows, %d rows\n", n, n * 24);
  for (int mode = 0; mode < 6; mode++) {
    const int grid = (int) (((size_t) n * (mode == 4 ? 96 : 72) + 255) / 256);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord (e0);
      switch (mode) {
        case 0: k < 0 ><<< grid, 256 >>> (img, d_win, n, W, out); break;
        case 1: k < 1 ><<< grid, 256 >>> (img, d_win, n, W, out); break;
        case 2: k < 2 ><<< grid, 256 >>> (img, d_win, n, W, out); break;
        case 3: k < 3 ><<< grid, 256 >>> (img, d_win, n, W, out); break;
        case 4: k < 4 ><<< grid, 256 >>> (img, d_win, n, W, out); break;
        case 5: k < 5 ><<< grid, 256 >>> (img, d_win, n, W, out); break;
      }
      hipEventRecord (e1); hipEventSynchronize (e1);
      hipEventElapsedTime (&ms, e0, e1);
    }
    printf ("%-44s %8.3f ms  %7.3f Gwindows/s  %6.1f Grows/s\n", names[mode], ms, n / ms / 1e6, n * 24.0 / ms / 1e6);
  }
  return 0;
}
