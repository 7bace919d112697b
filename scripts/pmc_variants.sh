#!/bin/bash
# Counter passes (one group per run, never beside a trace domain) over two of bench.py's decode variants -- full pel on the
# headline's blocks and the reference encoder's default picture (GPU box, from the repo root) -> gpurun_out/vpmc_<leg>_<pass>/
set -e
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
pass () {  # leg pass "counters" args...
  leg=$1; p=$2; ctrs=$3; shift; shift; shift
  rocprofv3 --pmc $ctrs --output-format csv -d $repo/gpurun_out/vpmc_${leg}_$p -o run -- python3 $repo/scripts/variant_run.py "$@" > $repo/gpurun_out/vpmc_${leg}_$p.log 2>&1
}
for leg in fullpel encdef headline; do
  case $leg in
    fullpel) args="prec=0 check=0 queues=1 steps=6";;
    encdef) args="xblen=32 xbsep=16 prec=0 check=0 queues=1 steps=6";;
    headline) args="check=0 queues=1 steps=6";;
  esac
  pass $leg fetch "FETCH_SIZE" $args
  pass $leg write "WRITE_SIZE" $args
  pass $leg busy "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum" $args
  pass $leg tcc "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD" $args
  cd $repo
  echo "== $leg"; python3 scripts/pmc_sum.py gpurun_out/vpmc_${leg}_fetch gpurun_out/vpmc_${leg}_write gpurun_out/vpmc_${leg}_busy gpurun_out/vpmc_${leg}_tcc | grep obmc
  cd /tmp
done
