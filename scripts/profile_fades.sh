#!/bin/bash
# rocprofv3 kernel stats of fades (picture weights 3, 5 / 2^3) on block sets other than the headline's (r06): gpurun_out/var_<name>/
set -e
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
run () {
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/var_$name -o run -- python3 $repo/scripts/variant_run.py "$@" > $repo/gpurun_out/var_$name.log 2>&1
  f=$(find $repo/gpurun_out/var_$name -name '*kernel_stats.csv' | head -1)
  echo "== $name"; grep '^{' $repo/gpurun_out/var_$name.log | cut -c1-200
  python3 - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-70s calls %5s avg %9.1f ns" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])))
PY
}
run fade3216fp xblen=32 xbsep=16 prec=0 weights=3,5,3 check=1 queues=1
run fade2416 xblen=24 xbsep=16 weights=3,5,3 check=1 queues=1
run fade168e xblen=16 xbsep=8 prec=3 weights=3,5,3 check=1 queues=1
