// Microbenchmark: the window gather of the OBMC row kernel (a wave = 64 block rows, lane = row, one
// dword-aligned 16-byte load per needed plane) under candidate half-pel plane layouts, to price a
// smaller footprint before building it:
//   MODE 1  r03: 32-byte chunks advancing 16 columns (every column twice), 4 rows per 128-byte line
//   MODE 4  64-byte chunks advancing 48 columns (1.33 x), 4 rows per 256 bytes (2 rows per line)
//   MODE 5  32-byte chunks, no overlap (1 x): a second load from the next chunk for every tap
//   MODE 6  as 5, the second load only for the runs that cross (38 %), as the real kernel would sort them
//   hipcc --offload-arch=gfx950 -O3 scripts/hp_layout_bench.hip -o build/hp_layout_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define G __attribute__ ((address_space (1)))
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
typedef u32x4 u32x4_a4 __attribute__ ((aligned (4)));

struct Win { int x, y, phases, ref; };

template < int MODE >
__global__ __launch_bounds__ (256) void gather_kernel (const uint8_t * img, const Win * win, int nitems, int W, int H, uint32_t * out)
{
  const int item = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  if (item < nitems) {
    const Win w = win[item / 12];
    const int row = item % 12;
    const int adv = MODE == 1 ? 16 : MODE == 4 ? 48 : 32, chunk = MODE == 4 ? 64 : 32;
    const int cpr = W / adv + 2;
    const size_t plane_bytes = (size_t) cpr * chunk * 4 * (H / 4);
    const uint8_t *base = img + (size_t) w.ref * 4 * plane_bytes;
    const int rx = w.phases & 1, ry = (w.phases >> 1) & 1;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int px = p & 1, py = p >> 1;
      if (!((!px || rx) && (!py || ry)))
        continue;
      const int X = w.x + px, Y = w.y + py;
      const int plane = (X & 1) + 2 * (Y & 1), x = X >> 1, y = (Y >> 1) + row;
      const int c = x / adv, o = x - c * adv;
      const size_t a = (size_t) plane * plane_bytes + ((size_t) (y >> 2) * cpr + c) * (chunk * 4) + (y & 3) * chunk + o;
      const u32x4 q = *(const G u32x4_a4 *) (base + (a & ~(size_t) 3));
      acc ^= q.x ^ q.y ^ q.z ^ q.w;
      if (MODE == 5 || (MODE == 6 && o + 13 > 32)) {
        const size_t b = (size_t) plane * plane_bytes + ((size_t) (y >> 2) * cpr + c + 1) * (chunk * 4) + (y & 3) * chunk;
        const u32x4 r = *(const G u32x4_a4 *) (base + b);
        acc ^= r.x ^ r.y ^ r.z ^ r.w;
      }
    }
  }
  if (acc == 0x12345678u)
    out[item & 1023] = acc;
}

int main ()
{
  uint32_t *out; (void) hipMalloc (&out, 4096);
  hipEvent_t e0, e1; (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
  const int W = 3840, H = 2160;
  const size_t bytes = (size_t) 2 * 4 * (W / 16 + 2) * 128 * (H / 4) + (1 << 20);
  uint8_t *img; (void) hipMalloc (&img, bytes);
  (void) hipMemset (img, 1, bytes);
  std::vector < Win > win;
  uint32_t s = 12345;
  for (int pic = 0; pic < 6; pic++)
    for (int ty = 0; ty < 66; ty++)
      for (int tx = 0; tx < 29; tx++) {
        const size_t t0 = win.size ();
        for (int by = 0; by < 4; by++)
          for (int bx = 0; bx < 16; bx++) {
            s = s * 1664525u + 1013904223u; int dx = (int) ((s >> 8) % 65) - 32;
            s = s * 1664525u + 1013904223u; int dy = (int) ((s >> 8) % 65) - 32;
            s = s * 1664525u + 1013904223u; int ref = (s >> 12) & 1;
            s = s * 1664525u + 1013904223u; int ph = (s >> 12) & 3;
            Win w; w.x = 64 + tx * 256 + bx * 16 + dx; w.y = 40 + ty * 64 + by * 16 + dy; w.phases = ph; w.ref = ref;
            win.push_back (w);
          }
        for (int i = 63; i > 0; i--) {
          s = s * 1664525u + 1013904223u;
          int j = (s >> 8) % (i + 1);
          Win tmp = win[t0 + i]; win[t0 + i] = win[t0 + j]; win[t0 + j] = tmp;
        }
      }
  const int n = (int) win.size (), nitems = n * 12;
  Win *d_win; (void) hipMalloc (&d_win, (size_t) n * sizeof (Win));
  (void) hipMemcpy (d_win, win.data (), (size_t) n * sizeof (Win), hipMemcpyHostToDevice);
  const int modes[] = { 1, 4, 5, 6 };
  const char *names[] = { "r03: 32-byte chunks advancing 16 (2 x), 4 rows per line", "64-byte chunks advancing 48 (1.33 x), 2 rows per line",
    "32-byte chunks, no overlap (1 x), always a second load", "32-byte chunks, no overlap (1 x), second load where the run crosses" };
  printf ("%d windows (12 rows each, lane = row)\n", n);
  for (int k = 0; k < 4; k++) {
    const int grid = (nitems + 255) / 256;
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      (void) hipEventRecord (e0);
      switch (modes[k]) {
        case 1: gather_kernel < 1 ><<< grid, 256 >>> (img, d_win, nitems, W, H, out); break;
        case 4: gather_kernel < 4 ><<< grid, 256 >>> (img, d_win, nitems, W, H, out); break;
        case 5: gather_kernel < 5 ><<< grid, 256 >>> (img, d_win, nitems, W, H, out); break;
        case 6: gather_kernel < 6 ><<< grid, 256 >>> (img, d_win, nitems, W, H, out); break;
      }
      (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
      (void) hipEventElapsedTime (&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf ("%-70s %8.3f ms  %7.3f Gwindows/s\n", names[k], best, n / best / 1e6);
  }
  return 0;
}
