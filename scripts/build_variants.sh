#!/bin/bash
# kernel-variant libraries for in-process A/B runs: bash scripts/build_variants.sh <file.hip> <MACRO> v1 v2 ...  -> ab/libclt_hip_<MACRO>_<v>.so
set -e
cd "$(dirname "$0")/../crosslayer_transcoder_amd/csrc"
src=$1; macro=$2; shift 2
make -s -j8
mkdir -p ../../ab
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -D${macro}=$v -c $src -o ../../ab/${src%.hip}_${macro}_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../ab/${src%.hip}_${macro}_$v.o -o ../../ab/libclt_hip_${macro}_$v.so
done
