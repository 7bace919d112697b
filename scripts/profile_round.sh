#!/bin/bash
# Round profile artefacts (run on the GPU box from the repo root; ROUND=r04 by default):
#   gpurun_out/kt/     rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/pmc_*   FETCH_SIZE / WRITE_SIZE, cache and instruction counters, one pass each
#                      (counters are never combined with a trace domain)
# Copy the summaries into profiles/ afterwards (profiles/${ROUND}_*).
set -e
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/kt -o run -- python3 $repo/bench.py --headline-only > $repo/gpurun_out/kt.log 2>&1
cd $repo
scripts/pmc_pass.sh fetch "FETCH_SIZE"
scripts/pmc_pass.sh write "WRITE_SIZE"
scripts/pmc_pass.sh tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"
scripts/pmc_pass.sh inst "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"
scripts/pmc_pass.sh busy "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum"
python3 scripts/pmc_sum.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_tcc gpurun_out/pmc_inst gpurun_out/pmc_busy > gpurun_out/pmc_summary.txt
cat gpurun_out/kt/run_kernel_stats.csv
