#!/bin/bash
# A/B build of ONE source file: build_variant_one.sh NAME FILE.hip [extra hipcc flags] -> turbo-range-coder_amd/build/ab/libNAME.so
# (the other objects come from turbo-range-coder_amd/build/: run `make` there first; select at run time with TRC_LIB=...)
set -e
cd "$(dirname "$0")/.."
NAME=$1; FILE=$2; shift 2
PKG=turbo-range-coder_amd
OUT=$PKG/build/ab; mkdir -p $OUT/$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Iinclude "$@" -c $PKG/csrc/$FILE -o $OUT/$NAME/${FILE%.hip}.o
OBJS=$(ls $PKG/build/*.o | grep -v "/${FILE%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$NAME.so $OUT/$NAME/${FILE%.hip}.o $OBJS
echo built $OUT/lib$NAME.so
