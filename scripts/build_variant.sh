#!/bin/bash
# scripts/build_variant.sh NAME "-DFLAG=..": a second build of libschro_hip.so (build/libschro_hip_NAME.so)
# with extra compiler flags, for A/B runs on the GPU box: SCHRO_HIP_LIB=build/libschro_hip_NAME.so
set -e
name=$1; shift
cd "$(dirname "$0")/../schroedinger_amd/csrc"
mkdir -p ../../build/v_$name
for f in context.cpp plane_iiwt.cpp plane_frameops.cpp plane_lowdelay.cpp plane_obmc.cpp iiwt_pack.cpp frame.cpp scheduler.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. "$@" -x hip -c $f -o ../../build/v_$name/${f%.*}.o &
done
for f in iiwt.hip iiwt_reg.hip iiwt_haar.hip frameops.hip obmc.hip obmc_row.hip obmc_strip.hip lowdelay.hip dequant.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. "$@" -c $f -o ../../build/v_$name/${f%.*}.o &
done
for j in $(jobs -p); do wait $j || { echo "build_variant: compile failed"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libschro_hip_$name.so ../../build/v_$name/*.o
echo build/libschro_hip_$name.so
