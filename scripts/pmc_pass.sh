#!/bin/bash
# One rocprofv3 counter pass per argument group over a short bench.py run (GPU box only).
# usage: scripts/pmc_pass.sh NAME "CTR1 CTR2 ..."   -> gpurun_out/pmc_NAME/
set -e
name=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $1 --output-format csv -d $repo/gpurun_out/pmc_$name -o run -- python3 $repo/bench.py --steps 4 --warmup 2 --queues 1 --headline-only > $repo/gpurun_out/pmc_$name.log 2>&1
