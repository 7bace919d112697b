#!/bin/bash
# scripts/ab_env.sh OUT VAR VALUE...: bench.py --headline-only with the EXPERIMENTS library once per value of one of its
# environment switches (VALUE "-": the variable unset), twice round, same box -> gpurun_out/OUT.txt
out=gpurun_out/$1.txt; var=$2; shift; shift
: > $out
export SCHRO_HIP_LIB=$PWD/schroedinger_amd/libschro_hip_exp.so
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  python3 bench.py --headline-only --steps 60 --warmup 10 ${AB_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['kernels']
print('%-28s step %.4f  obmc %.4f  iiwt_finest %.4f  coarse %.4f  upsample %.4f' % ('$var=$v', d['ms_per_step'], k['obmc']['ms_per_step'], k['iiwt_finest']['ms_per_step'], k['iiwt_coarse']['ms_per_step'], k['upsample']['ms_per_step']))" >> $out || echo "$var=$v FAILED" >> $out
done
done
cat $out
