#!/bin/bash
# instruction mix and wave-cycle accounting of the headline's kernels (GPU box): counter passes only, one per group
set -e
scripts/pmc_pass.sh mix1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES"
scripts/pmc_pass.sh mix2 "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_ANY"
python3 scripts/pmc_sum.py gpurun_out/pmc_mix1 gpurun_out/pmc_mix2 > gpurun_out/pmc_mix.txt
grep obmc gpurun_out/pmc_mix.txt
