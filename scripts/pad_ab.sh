out=gpurun_out/pad_ab.txt; : > $out
export SCHRO_HIP_LIB=$PWD/schroedinger_amd/libschro_hip_exp.so
for rep in 1 2; do
for pad in 0 2048 4096 8192; do
  SCHRO_HIP_OBMC_LDS_PAD=$pad python3 bench.py --headline-only --steps 60 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['kernels']
print('pad %-6s step %.4f  obmc %.4f  iiwt_finest %.4f  coarse %.4f  upsample %.4f' % ('$pad', d['ms_per_step'], k['obmc']['ms_per_step'], k['iiwt_finest']['ms_per_step'], k['iiwt_coarse']['ms_per_step'], k['upsample']['ms_per_step']))" >> $out
done
done
cat $out
