#!/bin/bash
# rocprofv3 kernel stats of bench.py's decode variants (GPU box): gpurun_out/var_<name>/
set -e
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
run () {
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/var_$name -o run -- python3 $repo/scripts/variant_run.py "$@" > $repo/gpurun_out/var_$name.log 2>&1
  f=$(find $repo/gpurun_out/var_$name -name '*kernel_stats.csv' | head -1)
  echo "== $name"; cat $repo/gpurun_out/var_$name.log | tail -1 | cut -c1-300
  python3 - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-70s calls %5s avg %9.1f ns" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])))
PY
}
# (queues=1: one batch in flight, so a kernel's average duration is its own)
run fullpel prec=0 check=0 queues=1
run eighth prec=3 check=0 queues=1
run b2416 xblen=24 xbsep=16 check=0 queues=1
run p1080 w=1920 h=1080 check=0 queues=1
run headline check=0 queues=1
run encdef xblen=32 xbsep=16 prec=0 check=0 queues=1
run fade weights=3,5,3 check=0 queues=1
