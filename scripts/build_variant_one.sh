#!/bin/bash
# scripts/build_variant_one.sh NAME FILE "-DFLAG=..." ...: build/libschro_hip_NAME.so = the product library's objects
# (schroedinger_amd/csrc/*.o, built by make) with FILE (a .hip or .cpp of csrc/) recompiled with the extra flags --
# an A/B build in the time of one file.  Use: SCHRO_HIP_LIB=build/libschro_hip_NAME.so
set -e
name=$1; file=$2; shift; shift
cd "$(dirname "$0")/../schroedinger_amd/csrc"
mkdir -p ../../build/v_$name
x=""; case $file in *.cpp) x="-x hip";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. "$@" $x -c $file -o ../../build/v_$name/${file%.*}.o
objs=""
for f in $(sed -n 's/^SRCS = //p' Makefile); do
  f=${f%.*}
  if [ "$f" = "${file%.*}" ]; then objs="$objs ../../build/v_$name/$f.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libschro_hip_$name.so $objs
echo build/libschro_hip_$name.so
