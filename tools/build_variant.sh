#!/bin/bash
# Kernel A/B builds: tools/build_variant.sh <tag> [-DFLAG ...]  -> build/libsdirt_dp_<tag>.so
# (same sources as the product library plus the given -D switches; used with
#  SDIRT_AMD_LIB=build/libsdirt_dp_<tag>.so python tools/kbench.py)
set -e
TAG=$1; shift
cd "$(dirname "$0")/../sdirt_amd/csrc"
mkdir -p ../../build
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS "$@" -c sdirt_dp.hip -o ../../build/sdirt_dp_$TAG.o
[ -f sdirt_mlp.o ] || make -s sdirt_mlp.o sdirt_dfdp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared ../../build/sdirt_dp_$TAG.o sdirt_mlp.o sdirt_dfdp.o -o ../../build/libsdirt_dp_$TAG.so
rm -f ../../build/sdirt_dp_$TAG.o
echo built build/libsdirt_dp_$TAG.so
