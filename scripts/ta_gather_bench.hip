// microbenchmark: cost of gathering 24-byte row pieces with different lane shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define G __attribute__ ((address_space (1)))
typedef uint32_t u32x2 __attribute__ ((ext_vector_type (2)));
typedef uint32_t u32x3 __attribute__ ((ext_vector_type (3)));
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
typedef uint32_t u32_u __attribute__ ((aligned (1)));
typedef u32x2 u32x2_u __attribute__ ((aligned (1)));
typedef u32x3 u32x3_u __attribute__ ((aligned (1)));
typedef u32x4 u32x4_u __attribute__ ((aligned (1)));

// each "row piece" k: address = img + rowidx[k] * stride + xoff[k]
// MODE: lanes per piece LP, bytes per lane
template < int MODE >
__global__ __launch_bounds__ (256) void k (const uint8_t * img, const int *offs, int npieces, int iters, uint32_t * out)
{
  constexpr int LP = MODE == 0 ? 3 : MODE == 1 ? 4 : MODE == 2 ? 2 : MODE == 3 ? 1 : MODE == 4 ? 6 : MODE == 5 ? 2 : MODE == 6 ? 1 : 2;
  constexpr int BL = MODE == 0 ? 8 : MODE == 1 ? 8 : MODE == 2 ? 16 : MODE == 3 ? 16 : MODE == 4 ? 4 : MODE == 5 ? 12 : MODE == 6 ? 12 : 16;
  const int tid = blockIdx.x * 256 + threadIdx.x;
  const int lane_in = tid % LP;
  int piece = tid / LP;
  const int stride_p = gridDim.x * 256 / LP;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    const int o = offs[piece % npieces] + lane_in * BL;
    const uint8_t *p = img + o;
    if (BL == 4) acc += *(const G u32_u *) p;
    else if (BL == 8) { u32x2 v = *(const G u32x2_u *) p; acc += v.x ^ v.y; }
    else if (BL == 12) { u32x3 v = *(const G u32x3_u *) p; acc += v.x ^ v.y ^ v.z;
      if (MODE == 6) { u32x3 w = *(const G u32x3_u *) (p + 12); acc += w.x ^ w.y ^ w.z; } }
    else { u32x4 v = *(const G u32x4_u *) p; acc += v.x ^ v.y ^ v.z ^ v.w;
      if (MODE == 3) { u32x2 w = *(const G u32x2_u *) (p + 16); acc += w.x ^ w.y; } }
    piece += stride_p;
  }
  out[tid] = acc;
}

int main ()
{
  const int W = 7680, H = 1024;         // 7.5 MB image: L2 resident per XCD mostly
  uint8_t *img; hipMalloc (&img, (size_t) W * H + 4096);
  hipMemset (img, 1, (size_t) W * H + 4096);
  const int NP = 1 << 20;
  std::vector < int >offs (NP);
  uint32_t s = 12345;
  // pieces come in runs of 12 consecutive-ish rows (a block): rows r, r+2, r+4..., random x
  for (int b = 0; b < NP / 12; b++) {
    s = s * 1664525u + 1013904223u; int y0 = (s >> 8) % (H - 32);
    s = s * 1664525u + 1013904223u; int x0 = (s >> 8) % (W - 64);
    for (int r = 0; r < 12; r++) offs[b * 12 + r] = (y0 + 2 * r) * W + x0;
  }
  for (int i = NP / 12 * 12; i < NP; i++) offs[i] = 0;
  int *d_offs; hipMalloc (&d_offs, NP * 4); hipMemcpy (d_offs, offs.data (), NP * 4, hipMemcpyHostToDevice);
  uint32_t *out; hipMalloc (&out, 4 * 256 * 8192);
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  const char *names[] = { "x2, 3 lanes/row (24 B)", "x2, 4 lanes/row (32 B)", "x4, 2 lanes/row (32 B)", "x4+x2, 1 lane/row (24 B)",
    "x1, 6 lanes/row (24 B)", "x3, 2 lanes/row (24 B)", "x3+x3, 1 lane/row (24 B)", "x4 2 lanes (dup)" };
  for (int mode = 0; mode < 7; mode++) {
    const int grid = 4096, iters = 64;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord (e0);
      switch (mode) {
        case 0: k < 0 ><<< grid, 256 >>> (img, d_offs, NP, iters, out); break;
        case 1: k < 1 ><<< grid, 256 >>> (img, d_offs, NP, iters, out); break;
        case 2: k < 2 ><<< grid, 256 >>> (img, d_offs, NP, iters, out); break;
        case 3: k < 3 ><<< grid, 256 >>> (img, d_offs, NP, iters, out); break;
        case 4: k < 4 ><<< grid, 256 >>> (img, d_offs, NP, iters, out); break;
        case 5: k < 5 ><<< grid, 256 >>> (img, d_offs, NP, iters, out); break;
        case 6: k < 6 ><<< grid, 256 >>> (img, d_offs, NP, iters, out); break;
      }
      hipEventRecord (e1); hipEventSynchronize (e1);
    }
    float ms; hipEventElapsedTime (&ms, e0, e1);
    const int LPs[] = { 3, 4, 2, 1, 6, 2, 1 };
    double pieces = (double) grid * 256 / LPs[mode] * iters;
    printf ("%-28s %8.3f ms  %7.2f Gpieces/s  (%6.1f GB/s useful @24B)\n", names[mode], ms, pieces / ms / 1e6, pieces * 24 / ms / 1e6);
  }
  return 0;
}
