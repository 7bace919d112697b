// Microbenchmark: issue rate of the integer instructions the bit readers and the DC prediction are
// made of (per SIMD, by waves per SIMD), beside v_add_u32.
//   hipcc --offload-arch=gfx950 -O3 scripts/issue_rate_int.hip -o build/issue_rate_int
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template < int OP > __device__ __forceinline__ uint32_t
op (uint32_t a, uint32_t b, uint32_t c)
{
  uint32_t d;
  if constexpr (OP == 0)
    asm volatile ("v_add_u32 %0, %1, %2" : "=v" (d) : "v" (a), "v" (b));
  else if constexpr (OP == 1) {
    uint64_t q = ((uint64_t) a << 32) | b, r;
    asm volatile ("v_lshlrev_b64 %0, %1, %2" : "=v" (r) : "v" (c), "v" (q));
    d = (uint32_t) (r >> 32);
  } else if constexpr (OP == 2)
    asm volatile ("v_mul_hi_i32 %0, %1, %2" : "=v" (d) : "v" (a), "v" (b));
  else if constexpr (OP == 3)
    asm volatile ("v_mul_lo_u32 %0, %1, %2" : "=v" (d) : "v" (a), "v" (b));
  else if constexpr (OP == 4)
    asm volatile ("v_mad_u32_u24 %0, %1, %2, %3" : "=v" (d) : "v" (a), "v" (b), "v" (c));
  else if constexpr (OP == 5)
    asm volatile ("v_ffbh_u32 %0, %1" : "=v" (d) : "v" (a ^ b));
  else if constexpr (OP == 6)
    asm volatile ("v_bfe_u32 %0, %1, %2, 1" : "=v" (d) : "v" (a), "v" (b));
  else if constexpr (OP == 7)
    asm volatile ("v_add3_u32 %0, %1, %2, %3" : "=v" (d) : "v" (a), "v" (b), "v" (c));
  else if constexpr (OP == 8)
    asm volatile ("v_alignbit_b32 %0, %1, %2, %3" : "=v" (d) : "v" (a), "v" (b), "v" (c));
  else if constexpr (OP == 9)
    asm volatile ("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v" (d) : "v" (a ^ b));
  else
    asm volatile ("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v" (*(uint64_t *) &d) : "v" (a), "v" (b), "v" ((uint64_t) c) : "vcc");
  return d;
}

template < int OP >
__global__ __launch_bounds__ (256) void issue_kernel (uint32_t * out, unsigned long long *cycles, int iters)
{
  uint32_t r[16];
#pragma unroll
  for (int i = 0; i < 16; i++)
    r[i] = threadIdx.x * 2654435761u + i * 40503u;
  const uint32_t c = 3u + (threadIdx.x & 1);
  const uint64_t t0 = __builtin_amdgcn_s_memtime ();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++)
      r[i] = op < OP > (r[i], r[(i + 5) & 15], c);
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

int main ()
{
  hipDeviceProp_t prop;
  (void) hipGetDeviceProperties (&prop, 0);
  const int cus = prop.multiProcessorCount;
  uint32_t *out; (void) hipMalloc (&out, 4096);
  unsigned long long *cyc; (void) hipMalloc (&cyc, 8 * 65536);
  const char *opn[] = { "v_add_u32", "v_lshlrev_b64", "v_mul_hi_i32", "v_mul_lo_u32", "v_mad_u32_u24", "v_ffbh_u32 (+ xor)", "v_bfe_u32",
    "v_add3_u32", "v_alignbit_b32", "v_mov_b32_dpp wave_shr (+ xor)" };
  const int iters = 2048;
  printf ("%d CUs.  s_memtime ticks per wave-instruction per SIMD = in-kernel ticks x waves per SIMD / instructions\n", cus);
  for (int o = 0; o < 10; o++) {
    printf ("%-32s", opn[o]);
    for (int wps = 1; wps <= 8; wps *= 2) {
      const int grid = cus * wps;
      for (int rep = 0; rep < 2; rep++) {
        switch (o) {
#define CASE(n) case n: issue_kernel < n ><<< grid, 256 >>> (out, cyc, iters); break;
          CASE (0) CASE (1) CASE (2) CASE (3) CASE (4) CASE (5) CASE (6) CASE (7) CASE (8) CASE (9)
#undef CASE
        }
        (void) hipDeviceSynchronize ();
      }
      std::vector < unsigned long long >h (grid);
      (void) hipMemcpy (h.data (), cyc, 8 * (size_t) grid, hipMemcpyDeviceToHost);
      double mean = 0;
      for (auto v : h) mean += (double) v;
      mean /= grid;
      printf ("  w/SIMD %d: %5.2f", wps, mean / (16.0 * iters * wps));
    }
    printf ("\n");
  }
  return 0;
}
