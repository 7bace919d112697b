// scripts/store_pattern_bench.hip -- would a register form of the upsample (a lane walks DOWN its columns, one row of all four
// planes per step) write the half-pel planes as fast as the LDS form does?  The planes' lines are 4 rows x 32 bytes of one
// plane of one chunk (schro_hip_internal.h).  A: the LDS form's stores -- a wave instruction = 4 rows x 16 lanes x 16 bytes =
// eight WHOLE lines.  B: a lane per 16-byte piece of a row, the rows one after the other -- a wave instruction = 64 pieces of
// ONE row = a quarter of each of 32 lines, the other three quarters in the next three instructions.  Same bytes, same layout.
//   hipcc --offload-arch=gfx950 -O2 -o build/store_pattern_bench scripts/store_pattern_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));

// bands of 4 rows; a band = chunks x 512 bytes; a chunk = 4 planes x (4 rows x 32 bytes)
__global__ void form_a (char *dst, int chunks, int bands)
{
  // a wave: 8 chunks of one band; lane = row (2 bits) | piece (4 bits: chunk 3 bits, half 1 bit)
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int per_band = chunks / 8, band = wave / per_band, c0 = (wave % per_band) * 8;
  if (band >= bands)
    return;
  const int r = lane >> 4, g = lane & 15;
  char *p = dst + (size_t) band * chunks * 512 + (size_t) (c0 + (g >> 1)) * 512 + r * 32 + (g & 1) * 16;
  const u32x4 v = { (uint32_t) wave, (uint32_t) lane, 3u, 4u };
#pragma unroll
  for (int pl = 0; pl < 4; pl++)
    *(u32x4 *) (p + pl * 128) = v;
}

__global__ void form_b (char *dst, int chunks, int bands, int rows_per_wave)
{
  // a wave: 32 chunks wide, rows_per_wave rows down; lane = piece (chunk 5 bits, half 1 bit)
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int per_row = chunks / 32, strip = wave / per_row, c0 = (wave % per_row) * 32;
  const int y0 = strip * rows_per_wave;
  if (y0 >= bands * 4)
    return;
  const u32x4 v = { (uint32_t) wave, (uint32_t) lane, 3u, 4u };
  for (int y = y0; y < y0 + rows_per_wave; y++) {
    char *p = dst + (size_t) (y >> 2) * chunks * 512 + (size_t) (c0 + (lane >> 1)) * 512 + (y & 3) * 32 + (lane & 1) * 16;
#pragma unroll
    for (int pl = 0; pl < 4; pl++)
      *(u32x4 *) (p + pl * 128) = v;
  }
}

int main ()
{
  const int chunks = 256, bands = 544 * 2;     // 3840 + 2 x 32 pixels + the spare chunk; 2 x 2160 rows: 2 x 71 MB
  const size_t bytes = (size_t) chunks * 512 * bands;
  char *d;
  hipMalloc ((void **) &d, bytes);
  hipEvent_t a, b;
  hipEventCreate (&a);
  hipEventCreate (&b);
  for (int form = 0; form < 4; form++)
    for (int rep = 0; rep < 3; rep++) {
      hipMemset (d, 0, bytes);
      hipDeviceSynchronize ();
      hipEventRecord (a, 0);
      const int rows[4] = { 0, 8, 16, 32 };
      if (form == 0) {
        const int waves = bands * (chunks / 8);
        form_a <<< (waves * 64 + 255) / 256, 256 >>> (d, chunks, bands);
      } else {
        const int waves = (bands * 4 / rows[form]) * (chunks / 32);
        form_b <<< (waves * 64 + 255) / 256, 256 >>> (d, chunks, bands, rows[form]);
      }
      hipEventRecord (b, 0);
      hipEventSynchronize (b);
      float ms = 0;
      hipEventElapsedTime (&ms, a, b);
      printf ("form %s%d: %.1f us  %.2f TB/s\n", form ? "B rows " : "A ", rows[form], ms * 1e3f, bytes / (ms * 1e-3) / 1e12);
    }
  return 0;
}
