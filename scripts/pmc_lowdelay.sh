#!/bin/bash
# Counter passes over the config-5 leg alone (GPU box, from the repo root) -> gpurun_out/ldpmc_<pass>/
set -e
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
for pass in "inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" \
            "busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
            "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  p=${pass%% *}; ctrs=${pass#* }
  rocprofv3 --pmc $ctrs --output-format csv -d $repo/gpurun_out/ldpmc_$p -o run -- python3 $repo/scripts/only.py lowdelay_8k > $repo/gpurun_out/ldpmc_$p.log 2>&1
done
cd $repo
python3 scripts/pmc_sum.py gpurun_out/ldpmc_inst gpurun_out/ldpmc_busy gpurun_out/ldpmc_fetch gpurun_out/ldpmc_write > gpurun_out/ldpmc_summary.txt
