#!/bin/bash
# rocprofv3 kernel stats of the block lengths between the presets (r06: 20 and 28 as two segments): gpurun_out/var_<name>/
set -e
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
run () {
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/var_$name -o run -- python3 $repo/scripts/variant_run.py "$@" > $repo/gpurun_out/var_$name.log 2>&1
  f=$(find $repo/gpurun_out/var_$name -name '*kernel_stats.csv' | head -1)
  echo "== $name"; cat $repo/gpurun_out/var_$name.log | tail -1 | cut -c1-400
  python3 - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-70s calls %5s avg %9.1f ns" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])))
PY
}
run b2012 xblen=20 xbsep=12 check=1 queues=1
run b2816 xblen=28 xbsep=16 check=1 queues=1
run b2816fp xblen=28 xbsep=16 prec=0 check=1 queues=1
