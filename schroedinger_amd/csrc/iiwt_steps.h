// iiwt_steps.h -- the seven Dirac synthesis filters as lifting-step tables, shared by
// the LDS tile kernel (iiwt.hip) and the register kernel (iiwt_reg.hip).
#pragma once

namespace schro {

enum { K_ADD2_22, K_AVG11, K_MAS4, K_HAAR_HALF, K_HAAR_FULL, K_MAS8, K_MAS2 };

struct Step {
  int target;                   // 0: A (even / low half) updated from B, 1: B from A
  int kind;
  int off;                      // first neighbour index relative to i
  int sign;
  int c, rnd, sh;
};

// Synthesis step lists: schro_synth_ext_desl93 :1466, _53 :1542, _135 :1616,
// haar :1697-1764, _fidelity :1768, _daub97 :1894 (same table as
// oracle/oracle_wavelet.c, which is checked against the reference's kernels).
__host__ __device__ constexpr int
filter_nsteps (int f)
{
  return f == 6 ? 4 : 2;
}

__host__ __device__ constexpr Step
filter_step (int f, int k)
{
  switch (f) {
    case 0:
      return k == 0 ? Step {0, K_ADD2_22, -1, -1, 0, 2, 2}
                    : Step {1, K_MAS4, -1, +1, 0, 8, 4};
    case 1:
      return k == 0 ? Step {0, K_ADD2_22, -1, -1, 0, 2, 2}
                    : Step {1, K_AVG11, 0, +1, 0, 1, 1};
    case 2:
      return k == 0 ? Step {0, K_MAS4, -2, -1, 0, 16, 5}
                    : Step {1, K_MAS4, -1, +1, 0, 8, 4};
    case 3:
    case 4:
      return k == 0 ? Step {0, K_HAAR_HALF, 0, -1, 0, 1, 1}
                    : Step {1, K_HAAR_FULL, 0, +1, 0, 0, 0};
    case 5:
      return k == 0 ? Step {1, K_MAS8, -3, +1, 0, 128, 8}
                    : Step {0, K_MAS8, -4, +1, 1, 127, 8};
    default:
      return k == 0 ? Step {0, K_MAS2, -1, -1, 1817, 2048, 12}
           : k == 1 ? Step {1, K_MAS2, 0, -1, 3616, 2048, 12}
           : k == 2 ? Step {0, K_MAS2, -1, +1, 217, 2048, 12}
                    : Step {1, K_MAS2, 0, +1, 6497, 2048, 12};
  }
}

// lifting halo in sub-band samples (both directions)
__host__ __device__ constexpr int
filter_halo (int f)
{
  return f == 0 ? 2 : f == 1 ? 1 : f == 2 ? 3 : f == 5 ? 7 : f == 6 ? 2 : 0;
}

// 0 none, 1 wrapping (x+1)>>1 (orc_interleave2_rrshift1_*), 2 avgs(x,0)
// (orc_haar_synth_rrshift1_int_*)
__host__ __device__ constexpr int
filter_shift (int f)
{
  return (f == 3 || f == 5) ? 0 : (f == 4 ? 2 : 1);
}

__host__ __device__ constexpr int
kind_ntaps (int kind)
{
  return kind == K_MAS4 ? 4 : kind == K_MAS8 ? 8
       : (kind == K_HAAR_HALF || kind == K_HAAR_FULL) ? 1 : 2;
}

}                               // namespace schro
