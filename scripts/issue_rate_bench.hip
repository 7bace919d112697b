// Microbenchmark: what a SIMD of gfx950 issues per cycle for the byte-parallel vector
// instructions the OBMC passes are made of, by waves per SIMD; and the window gather of a
// plane-separated half-pel layout (32-byte chunks that overlap by 16, 4 rows per line) against
// the r02 layout (16-byte chunks, 8 rows of one parity per line).
//   hipcc --offload-arch=gfx950 -O3 scripts/issue_rate_bench.hip -o build/issue_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define G __attribute__ ((address_space (1)))
typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
typedef uint32_t u32x3 __attribute__ ((ext_vector_type (3)));
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
typedef u32x4 u32x4_a4 __attribute__ ((aligned (4)));
typedef u32x3 u32x3_a1 __attribute__ ((aligned (1)));
typedef unsigned short u16x2 __attribute__ ((ext_vector_type (2)));

// ---- 1. issue rate -------------------------------------------------------------------------
// OP 0 v_perm_b32, 1 v_lerp_u8, 2 v_alignbyte_b32, 3 v_bfi_b32, 4 v_pk_mul_lo_u16, 5 v_add_u32,
// 6 v_xor_b32, 7 v_and_or_b32, 8 mix (perm, lerp, alignbyte, bfi in turn)
template < int OP > __device__ __forceinline__ uint32_t
op (uint32_t a, uint32_t b, uint32_t c)
{
  if constexpr (OP == 0)
    return __builtin_amdgcn_perm (a, b, c);
  else if constexpr (OP == 1)
    return __builtin_amdgcn_lerp (a, b, c);
  else if constexpr (OP == 2)
    return __builtin_amdgcn_alignbyte (a, b, c);
  else if constexpr (OP == 3)
    return (a & c) | (b & ~c);
  else if constexpr (OP == 4)
    return __builtin_bit_cast (uint32_t, (u16x2) (__builtin_bit_cast (u16x2, a) * __builtin_bit_cast (u16x2, b)));
  else if constexpr (OP == 5)
    return a + b;
  else if constexpr (OP == 6)
    return a ^ b;
  else
    return (a & b) | c;
}

template < int OP >
__global__ __launch_bounds__ (256) void issue_kernel (uint32_t * out, unsigned long long *cycles, int iters)
{
  uint32_t r[16];
#pragma unroll
  for (int i = 0; i < 16; i++)
    r[i] = threadIdx.x * 2654435761u + i * 40503u;
  const uint32_t c = 0x01010101u + (threadIdx.x & 1);
  const uint64_t t0 = __builtin_amdgcn_s_memtime ();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if constexpr (OP == 8) {
        if ((i & 3) == 0) r[i] = op < 0 > (r[i], r[(i + 5) & 15], c);
        else if ((i & 3) == 1) r[i] = op < 1 > (r[i], r[(i + 5) & 15], c);
        else if ((i & 3) == 2) r[i] = op < 2 > (r[i], r[(i + 5) & 15], c & 3);
        else r[i] = op < 3 > (r[i], r[(i + 5) & 15], c);
      } else {
        r[i] = op < OP > (r[i], r[(i + 5) & 15], OP == 2 ? (c & 3) : c);
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime ();
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 16; i++)
    x ^= r[i];
  if (x == 0x12345u)
    out[0] = x;
  if (threadIdx.x == 0)
    cycles[blockIdx.x] = t1 - t0;
}

// ---- 2. window gather ----------------------------------------------------------------------
// A wave = 64 (block, row) items, as in the row kernel: 12 rows per block, lanes of a block read
// consecutive rows.  Windows: random vectors around a 16-sample grid.
//  MODE 0: r02 layout: per item 2 sample rows x 3 aligned 16-byte chunks (tile 16 B x 8 rows of one parity)
//  MODE 1: planar, 32-byte overlapping chunks x 4 rows: NP planes per window (1, 2 or 4 by the window's
//          quarter phases), one dword-aligned 16-byte load per plane
//  MODE 2: as 1, byte-aligned 12-byte loads
//  MODE 3: as 1 but always four loads (unused planes redirected to the first plane's address)
struct Win { int x, y, phases, ref; };
template < int MODE >
__global__ __launch_bounds__ (256) void gather_kernel (const uint8_t * img, const Win * win, int nitems, int W, int H, uint32_t * out)
{
  const int item = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  if (item < nitems) {
    const Win w = win[item / 12];
    const int row = item % 12;
    if (MODE == 0) {
      // half-pel image 2W x 2H, tiled by parity
      const size_t stride = 2 * (size_t) W;
      const uint8_t *base = img + (size_t) w.ref * stride * 2 * H;
#pragma unroll
      for (int v = 0; v < 2; v++) {
        const int y = w.y + 2 * row + v, x = w.x & ~15;
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const size_t a = ((size_t) (y >> 4) * 2 + (y & 1)) * 8 * stride + (size_t) ((x >> 4) + j) * 128 + ((y >> 1) & 7) * 16;
          const u32x4 q = *(const G u32x4 *) (base + a);
          acc ^= q.x ^ q.y ^ q.z ^ q.w;
        }
      }
    } else {
      // four planes of W x H, chunk c of a row = bytes [16 c, 16 c + 32), 4 rows per 128-byte line
      const int cpr = W / 16 + 1;
      const size_t plane_bytes = (size_t) cpr * 128 * (H / 4);
      const uint8_t *base = img + (size_t) w.ref * 4 * plane_bytes;
      const int rx = w.phases & 1, ry = (w.phases >> 1) & 1;
      const int hx = w.x, hy = w.y;
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const int px = p & 1, py = p >> 1;
        bool need = (!px || rx) && (!py || ry);
        int X = hx + px, Y = hy + py;
        if (MODE == 3 && !need) {
          X = hx;
          Y = hy;
          need = true;
        }
        if (!need)
          continue;
        const int plane = (X & 1) + 2 * (Y & 1), x = X >> 1, y = (Y >> 1) + row;
        const size_t a = (size_t) plane * plane_bytes + ((size_t) (y >> 2) * cpr + (x >> 4)) * 128 + (y & 3) * 32 + (x & 15);
        if (MODE == 2) {
          const u32x3 q = *(const G u32x3_a1 *) (base + a);
          acc ^= q.x ^ q.y ^ q.z;
        } else {
          const u32x4 q = *(const G u32x4_a4 *) (base + (a & ~(size_t) 3));
          acc ^= q.x ^ q.y ^ q.z ^ q.w;
        }
      }
    }
  }
  if (acc == 0x12345678u)
    out[item & 1023] = acc;
}

int main ()
{
  hipDeviceProp_t prop;
  (void) hipGetDeviceProperties (&prop, 0);
  const int cus = prop.multiProcessorCount;
  uint32_t *out; (void) hipMalloc (&out, 4096);
  unsigned long long *cyc; (void) hipMalloc (&cyc, 8 * 65536);
  hipEvent_t e0, e1; (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
  const char *opn[] = { "v_perm_b32", "v_lerp_u8", "v_alignbyte_b32", "v_bfi_b32", "v_pk_mul_lo_u16", "v_add_u32", "v_xor_b32", "v_and_or_b32", "mix perm/lerp/align/bfi" };
  const int iters = 2048;
  printf ("%d CUs.  cycles per wave-instruction per SIMD = in-kernel cycles x waves per SIMD / instructions\n", cus);
  for (int o = 0; o < 9; o++) {
    printf ("%-24s", opn[o]);
    for (int wps = 1; wps <= 8; wps *= 2) {
      const int grid = cus * wps;       // 256-thread workgroups: one wave per SIMD each
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        (void) hipEventRecord (e0);
        switch (o) {
#define CASE(n) case n: issue_kernel < n ><<< grid, 256 >>> (out, cyc, iters); break;
          CASE (0) CASE (1) CASE (2) CASE (3) CASE (4) CASE (5) CASE (6) CASE (7) CASE (8)
#undef CASE
        }
        (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
        (void) hipEventElapsedTime (&ms, e0, e1);
      }
      std::vector < unsigned long long >h (grid);
      (void) hipMemcpy (h.data (), cyc, 8 * (size_t) grid, hipMemcpyDeviceToHost);
      double mean = 0;
      for (auto v : h) mean += (double) v;
      mean /= grid;
      const double ninst = 16.0 * iters;
      printf ("  w/SIMD %d: %5.2f cyc/inst/SIMD (%6.4f ns)", wps, mean / (ninst * wps), ms * 1e6 / (ninst * wps));
    }
    printf ("\n");
  }

  // ---- gather ----
  const int W = 3840, H = 2160;
  const size_t bytes = (size_t) 2 * 4 * (W / 16 + 1) * 128 * (H / 4) + (size_t) 2 * 2 * W * 2 * H + (1 << 20);
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
  const char *names[] = { "r02 tiled 16x8 by parity, 2 rows x 3 chunks", "planar 32x4 overlapped, needed planes, 16 B dword-aligned",
    "planar 32x4 overlapped, needed planes, 12 B byte-aligned", "planar 32x4 overlapped, always 4 loads" };
  printf ("%d windows (12 rows each, lane = row)\n", n);
  for (int mode = 0; mode < 4; mode++) {
    const int grid = (nitems + 255) / 256;
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
      (void) hipEventRecord (e0);
      switch (mode) {
        case 0: gather_kernel < 0 ><<< grid, 256 >>> (img, d_win, nitems, W, H, out); break;
        case 1: gather_kernel < 1 ><<< grid, 256 >>> (img, d_win, nitems, W, H, out); break;
        case 2: gather_kernel < 2 ><<< grid, 256 >>> (img, d_win, nitems, W, H, out); break;
        case 3: gather_kernel < 3 ><<< grid, 256 >>> (img, d_win, nitems, W, H, out); break;
      }
      (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
      (void) hipEventElapsedTime (&ms, e0, e1);
    }
    printf ("%-60s %8.3f ms  %7.3f Gwindows/s\n", names[mode], ms, n / ms / 1e6);
  }
  return 0;
}
