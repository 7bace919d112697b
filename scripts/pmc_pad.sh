#!/bin/bash
# Counters of the headline's luma launch in the experiments library, with and without the line-aligned item layout
# (SCHRO_HIP_OBMC_PAD=1) -> gpurun_out/ppmc_<leg>_<pass>/
set -e
repo=$(pwd)
export SCHRO_HIP_LIB=$repo/schroedinger_amd/libschro_hip_exp.so
cd /tmp && export TMPDIR=/tmp
for leg in base pad; do
  if [ $leg = pad ]; then export SCHRO_HIP_OBMC_PAD=1; else unset SCHRO_HIP_OBMC_PAD; fi
  for p in busy:"GRBM_GUI_ACTIVE TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum" inst:"SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" tcc:"TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
    name=${p%%:*}; ctrs=${p#*:}
    rocprofv3 --pmc $ctrs --output-format csv -d $repo/gpurun_out/ppmc_${leg}_$name -o run -- python3 $repo/scripts/variant_run.py check=0 queues=1 steps=6 > $repo/gpurun_out/ppmc_${leg}_$name.log 2>&1
  done
  cd $repo
  echo "== $leg"; python3 scripts/pmc_sum.py gpurun_out/ppmc_${leg}_busy gpurun_out/ppmc_${leg}_inst gpurun_out/ppmc_${leg}_tcc | grep "p_3_1"
  cd /tmp
done
