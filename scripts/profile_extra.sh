#!/bin/bash
# Round profile artefacts of the legs outside the headline (GPU box, from the repo root):
#   gpurun_out/x_<leg>/   rocprofv3 --kernel-trace --stats
#   gpurun_out/xpmc_<leg>_<pass>/  counter passes (never combined with a trace domain)
set -e
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
for leg in "iiwt_1080p 8" "iiwt_2160p" "iiwt_s32_2160p" "lowdelay_8k"; do
  set -- $leg
  rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/x_$1 -o run -- python3 $repo/scripts/only.py $@ > $repo/gpurun_out/x_$1.log 2>&1
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"; do
    p=${pass%% *}; ctrs=${pass#* }
    rocprofv3 --pmc $ctrs --output-format csv -d $repo/gpurun_out/xpmc_$1_$p -o run -- python3 $repo/scripts/only.py $@ > $repo/gpurun_out/xpmc_$1_$p.log 2>&1
  done
done
cd $repo
for leg in iiwt_1080p iiwt_2160p iiwt_s32_2160p lowdelay_8k; do
  echo "== $leg kernel stats (calls, total ns, average ns, min ns, max ns)"
  python3 -c "
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-100s %6s %12s %12.0f %9s %9s' % (r['Name'][:100], r['Calls'], r['TotalDurationNs'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
" gpurun_out/x_$leg/run_kernel_stats.csv
  echo "== $leg counters"; python3 scripts/pmc_sum.py gpurun_out/xpmc_${leg}_fetch gpurun_out/xpmc_${leg}_write gpurun_out/xpmc_${leg}_inst
done > gpurun_out/extra_summary.txt
