#!/bin/bash
# Developer script: builds the library with extra compiler flags into var_libs/NAME/libtfrec_amd.so
# (select it with TFREC_AMD_LIB=...).  Usage: tools/build_variant.sh NAME "-DFLAG ..."
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
OUT=var_libs/$NAME; mkdir -p $OUT/obj
for f in api sort train eval sampler exchange; do
  EXTRA=""; [ $f = eval ] && EXTRA="-mllvm --amdgpu-mfma-vgpr-form -fno-honor-nans"
  hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-pass-failed $EXTRA "$@" -c tfrec_amd/csrc/$f.hip -o $OUT/obj/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OUT/obj/*.o -o $OUT/libtfrec_amd.so
rm -rf $OUT/obj
echo built $OUT/libtfrec_amd.so
