// Microbenchmark: how many cycles the texture-address / L1 path of a gfx950 CU takes per
// vector load instruction, by access shape -- width, alignment, and how the 64 lanes' addresses
// lie in the 128-byte lines.  Everything hits L1 (16 KB per CU); 8 loads in flight per wave, 4 or 8
// waves per SIMD.  What decided the r03 half-pel layout.
//   hipcc --offload-arch=gfx950 -O3 scripts/ta_rate_bench.hip -o build/ta_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// W: bytes per lane (4, 8, 12, 16); the lane's address = base + group * GS + (lane % LPG) * S + M
// where group = lane / LPG: LPG lanes S bytes apart, then the next group GS bytes further
template < int W >
__device__ __forceinline__ void
load (const uint8_t * p, uint32_t * d)
{
  if constexpr (W == 4)
    asm volatile ("global_load_dword %0, %1, off" : "=v" (d[0]) : "v" (p) : "memory");
  else if constexpr (W == 8)
    asm volatile ("global_load_dwordx2 %0, %1, off" : "=v" (*(uint64_t *) d) : "v" (p) : "memory");
  else if constexpr (W == 12) {
    typedef uint32_t u32x3 __attribute__ ((ext_vector_type (3)));
    asm volatile ("global_load_dwordx3 %0, %1, off" : "=v" (*(u32x3 *) d) : "v" (p) : "memory");
  } else {
    typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));
    asm volatile ("global_load_dwordx4 %0, %1, off" : "=v" (*(u32x4 *) d) : "v" (p) : "memory");
  }
}

template < int W >
__global__ __launch_bounds__ (256) void k (const uint8_t * buf, int S, int M, int LPG, int GS, int span, uint32_t * out, int iters, int perm, int active)
{
  // perm > 1 (r06): the addresses dealt to the lanes like cards to `perm` hands -- lanes that were neighbours are 64 / perm apart,
  // so lanes that share a 64-byte sector are no longer adjacent: does the unit still fetch the sector once?
  int lane = threadIdx.x & 63;
  if (perm > 1)
    lane = (lane % perm) * (64 / perm) + lane / perm;
  const uint8_t *base = buf + (lane / LPG) * GS + (lane % LPG) * S + M;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    __attribute__ ((aligned (16))) uint32_t d[8][4];
    // (active < 64, r06: only the first `active` lanes of the wave take part in the loads -- is a load's price its lanes' or its own?)
    if (active < 0 ? ((threadIdx.x & 63) % (unsigned) -active) == 0 : (int) (threadIdx.x & 63) < active) {
#pragma unroll
      for (int j = 0; j < 8; j++)
        load < W > (base + ((j * span) & 16383), d[j]);
    }
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; j++)
      acc ^= d[j][0];
  }
  if (acc == 0x12345u)
    out[0] = acc;
}

struct Pat { const char *name; int W, S, M, LPG, GS, perm, active; };

int main ()
{
  hipDeviceProp_t prop;
  (void) hipGetDeviceProperties (&prop, 0);
  const int cus = prop.multiProcessorCount;
  uint8_t *buf; (void) hipMalloc (&buf, 1 << 20); (void) hipMemset (buf, 1, 1 << 20);
  uint32_t *out; (void) hipMalloc (&out, 4096);
  hipEvent_t e0, e1; (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
  const Pat pats[] = {
    { "x1 coalesced (lanes 4 B apart)", 4, 4, 0, 64, 0 },
    { "x4 aligned, lanes 16 B apart (8 per line: r02 rows)", 16, 16, 0, 64, 0 },
    { "x4 aligned, 12 lanes 16 B apart, blocks 1 KB apart", 16, 16, 0, 12, 1024 },
    { "x4 aligned, lanes 32 B apart", 16, 32, 0, 64, 0 },
    { "x4 aligned, lanes 64 B apart", 16, 64, 0, 64, 0 },
    { "x4 aligned, lanes 128 B apart (a line each)", 16, 128, 0, 64, 0 },
    { "x3 dword-aligned, lanes 32 B apart", 12, 32, 4, 64, 0 },
    { "x3 byte-aligned (+1), lanes 32 B apart", 12, 32, 1, 64, 0 },
    { "x3 byte-aligned (+5), lanes 32 B apart", 12, 32, 5, 64, 0 },
    { "x3 byte-aligned (+17), lanes 32 B apart", 12, 32, 17, 64, 0 },
    { "x3 byte-aligned (+5), 12 lanes 32 B apart, blocks 1 KB apart", 12, 32, 5, 12, 1024 },
    { "x4 dword-aligned (+4), lanes 32 B apart", 16, 32, 4, 64, 0 },
    { "x4 byte-aligned (+5), lanes 32 B apart", 16, 32, 5, 64, 0 },
    { "x2 byte-aligned (+3), lanes 32 B apart", 8, 32, 3, 64, 0 },
    { "x2 aligned, lanes 32 B apart", 8, 32, 0, 64, 0 },
    { "x2 aligned, lanes 8 B apart", 8, 8, 0, 64, 0 },
    { "x2 byte-aligned (+3), lanes 16 B apart", 8, 16, 3, 64, 0 },
    { "x3 byte-aligned (+1), lanes 16 B apart", 12, 16, 1, 64, 0 },
    { "x3 dword-aligned, lanes 16 B apart", 12, 16, 4, 64, 0 },
    { "x3 dword-aligned, lanes 12 B apart (contiguous)", 12, 12, 0, 64, 0 },
    { "x1 byte-aligned (+1), lanes 32 B apart", 4, 32, 1, 64, 0 },
    { "x1 aligned, lanes 32 B apart", 4, 32, 0, 64, 0 },
    { "x1 aligned, lanes 16 B apart", 4, 16, 0, 64, 0 },
    // r06: which lanes may share a sector -- the same 64 addresses, neighbours dealt 2 / 4 / 16 / 32 lanes apart
    { "x4 aligned, lanes 32 B apart, sector mates 2 lanes apart", 16, 32, 0, 64, 0, 2 },
    { "x4 aligned, lanes 32 B apart, sector mates 4 lanes apart", 16, 32, 0, 64, 0, 4 },
    { "x4 aligned, lanes 32 B apart, sector mates 16 lanes apart", 16, 32, 0, 64, 0, 16 },
    { "x4 aligned, lanes 32 B apart, sector mates 32 lanes apart", 16, 32, 0, 64, 0, 32 },
    { "x4 aligned, lanes 16 B apart, sector mates 4 lanes apart", 16, 16, 0, 64, 0, 4 },
    { "x4 aligned, lanes 16 B apart, sector mates k, k + 32, k + 1, k + 33", 16, 16, 0, 64, 0, 32 },
    { "x4 dword-aligned (+4), lanes 32 B apart, mates 32 apart", 16, 32, 4, 64, 0, 32 },
    // rows of a plain plane: every lane its own sector (rows 4 KB apart would miss L1: 256 B apart here)
    { "x4 dword-aligned (+4), lanes 256 B apart", 16, 256, 4, 64, 0, 0 },
    { "x4 (+4), pairs 16 B apart in a sector, pairs 256 B apart", 16, 16, 4, 2, 256, 0 },
    { "x4 (+4), the same pairs, mates 32 lanes apart", 16, 16, 4, 2, 256, 32 },
    // r06: lanes switched off (EXEC) -- the first 32 / 16 / 4 lanes of every wave load, the rest do not
    { "x4 (+4), lanes 256 B apart, 32 of 64 lanes", 16, 256, 4, 64, 0, 0, 32 },
    { "x4 (+4), lanes 256 B apart, 16 of 64 lanes", 16, 256, 4, 64, 0, 0, 16 },
    { "x4 (+4), lanes 256 B apart, 4 of 64 lanes", 16, 256, 4, 64, 0, 0, 4 },
    { "x4 (+4), lanes 32 B apart, 32 of 64 lanes", 16, 32, 4, 64, 0, 0, 32 },
    { "x4 (+4), lanes 32 B apart, 16 of 64 lanes", 16, 32, 4, 64, 0, 0, 16 },
    { "x4 (+4), lanes 256 B apart, lanes 0, 4, 8 .. (one per quad)", 16, 256, 4, 64, 0, 0, -4 },
    { "x4 (+4), lanes 256 B apart, lanes 0, 2, 4 .. (two per quad)", 16, 256, 4, 64, 0, 0, -2 },
  };
  const int iters = 512;
  printf ("%d CUs; ns per load instruction per CU (all L1 hits), at 4 and 8 waves per SIMD\n", cus);
  for (const Pat & p : pats) {
    const int span = 4096;
    printf ("%-64s", p.name);
    for (int wps = 4; wps <= 8; wps *= 2) {
      const int grid = cus * wps;
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        (void) hipEventRecord (e0);
        switch (p.W) {
          case 4: k < 4 ><<< grid, 256 >>> (buf, p.S, p.M, p.LPG, p.GS, span, out, iters, p.perm, p.active ? p.active : 64); break;
          case 8: k < 8 ><<< grid, 256 >>> (buf, p.S, p.M, p.LPG, p.GS, span, out, iters, p.perm, p.active ? p.active : 64); break;
          case 12: k < 12 ><<< grid, 256 >>> (buf, p.S, p.M, p.LPG, p.GS, span, out, iters, p.perm, p.active ? p.active : 64); break;
          default: k < 16 ><<< grid, 256 >>> (buf, p.S, p.M, p.LPG, p.GS, span, out, iters, p.perm, p.active ? p.active : 64); break;
        }
        (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
        (void) hipEventElapsedTime (&ms, e0, e1);
      }
      // loads per CU: wps * 4 waves * iters * 8
      printf ("  %7.2f", ms * 1e6 / (wps * 4.0 * iters * 8));
    }
    printf ("\n");
  }
  return 0;
}
