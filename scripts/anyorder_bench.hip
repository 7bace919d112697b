// scripts/anyorder_bench.hip -- does hipExtAnyOrderLaunch let independent kernels of ONE stream run beside each other on gfx950?
// (hip_ext.h says the flag is not supported on GFX9xx boards.)  Eight launches of a kernel that occupies one wave per CU for
// ~20 us: in order they take 8 x 20 us, beside each other ~20.   hipcc --offload-arch=gfx950 -O2 -o build/anyorder_bench scripts/anyorder_bench.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin (unsigned long long ticks, unsigned *out)
{
  const unsigned long long t0 = __builtin_amdgcn_s_memtime ();
  while (__builtin_amdgcn_s_memtime () - t0 < ticks) { }
  if (threadIdx.x == 0 && blockIdx.x == 0)
    out[0] = 1;
}

int main ()
{
  hipStream_t s;
  hipStreamCreate (&s);
  unsigned *d;
  hipMalloc (&d, 64);
  hipEvent_t a, b;
  hipEventCreate (&a);
  hipEventCreate (&b);
  for (int flags = 0; flags < 2; flags++)
    for (int rep = 0; rep < 3; rep++) {
      hipStreamSynchronize (s);
      hipEventRecord (a, s);
      for (int k = 0; k < 8; k++)
        hipExtLaunchKernelGGL (spin, dim3 (256), dim3 (64), 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, 40000ull, d);
      hipEventRecord (b, s);
      hipStreamSynchronize (s);
      float ms = 0;
      hipEventElapsedTime (&ms, a, b);
      printf ("flags %d: 8 launches %.1f us\n", flags, ms * 1e3f);
    }
  return 0;
}
