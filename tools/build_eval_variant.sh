#!/bin/bash
# Developer script: rebuilds only eval.hip with extra flags and links it with the other objects of the last
# `make` into var_libs/NAME/libtfrec_amd.so.  Usage: tools/build_eval_variant.sh NAME -DFLAG ...
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
OUT=var_libs/$NAME; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-pass-failed \
  -mllvm --amdgpu-mfma-vgpr-form -fno-honor-nans "$@" -c tfrec_amd/csrc/eval.hip -o $OUT/eval.o
hipcc --offload-arch=gfx950 -shared -fPIC build/csrc/api.o build/csrc/sort.o build/csrc/train.o build/csrc/sampler.o build/csrc/exchange.o $OUT/eval.o -o $OUT/libtfrec_amd.so
rm -f $OUT/eval.o
echo built $OUT/libtfrec_amd.so
