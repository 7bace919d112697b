#!/bin/bash
# scripts/ab_libs.sh OUT NAME...: bench.py --headline-only once per library variant (build/libschro_hip_NAME.so; "product" =
# the in-tree library), same box, one after the other; step time and kernel classes per variant -> gpurun_out/OUT.txt
out=gpurun_out/$1.txt; shift
: > $out
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = product ]; then unset SCHRO_HIP_LIB; else export SCHRO_HIP_LIB=$PWD/build/libschro_hip_$n.so; fi
  python3 bench.py --headline-only --steps 60 --warmup 10 ${AB_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['kernels']
print('%-12s step %.4f  obmc %.4f  iiwt_finest %.4f  coarse %.4f  upsample %.4f' % ('$n', d['ms_per_step'], k['obmc']['ms_per_step'], k['iiwt_finest']['ms_per_step'], k['iiwt_coarse']['ms_per_step'], k['upsample']['ms_per_step']))" >> $out || echo "$n FAILED" >> $out
done
done
cat $out
