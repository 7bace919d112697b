#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline with ONE batch in flight (kernel durations undisturbed by the
# other queue's kernels: these are the averages bench.py's per-class HIP-event times agree with) -> gpurun_out/kt1/
set -e
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/kt1 -o run -- python3 $repo/bench.py --queues 1 --headline-only > $repo/gpurun_out/kt1.log 2>&1
cd $repo
cat gpurun_out/kt1/run_kernel_stats.csv
