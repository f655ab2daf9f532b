#!/bin/bash
# A/B builds for ablations: build_variant.sh NAME [extra hipcc flags] -> turbo-range-coder_amd/build/ab/libNAME.so
# (select at run time with TRC_LIB=...; travels to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
SRC=${TRC_VARIANT_SRC:-turbo-range-coder_amd/csrc}
OUT=turbo-range-coder_amd/build/ab; mkdir -p $OUT/$NAME
for f in $SRC/*.hip; do
  o=$OUT/$NAME/$(basename ${f%.hip}).o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Iinclude "$@" -c $f -o $o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$NAME.so $OUT/$NAME/*.o
echo built $OUT/lib$NAME.so
