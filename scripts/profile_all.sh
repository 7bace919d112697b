#!/bin/bash
# Everything a round's profiles/ needs, in one call on the GPU box (from the repo root): the default bench line and the driver's
# form, rocprofv3 kernel stats of the headline (two batches in flight / one), the counter passes, the config-5 leg's stats and
# counters.  Copy the summaries into profiles/ (profiles/rNN_*) afterwards.
set -e
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_steps20.json 2> gpurun_out/bench_steps20.err
scripts/profile_round.sh > gpurun_out/profile_round.log 2>&1
scripts/profile_one_batch.sh > gpurun_out/profile_one_batch.log 2>&1
scripts/pmc_obmc_mix.sh > gpurun_out/pmc_mix.log 2>&1
repo=$(pwd)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/x_lowdelay_8k -o run -- python3 $repo/scripts/only.py lowdelay_8k > $repo/gpurun_out/x_lowdelay_8k.log 2>&1)
scripts/pmc_lowdelay.sh > gpurun_out/pmc_lowdelay.log 2>&1
# r06: the transforms by themselves (plain 2160p: north_star's own figure; 1080p; s32) and the decode variants' kernels
scripts/profile_extra.sh > gpurun_out/profile_extra.log 2>&1
scripts/profile_variants.sh > gpurun_out/variants.txt 2>&1
echo done
