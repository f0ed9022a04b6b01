#!/bin/bash
# usage: scripts/build_variant.sh <name> <extra hipcc flags...> — a tuning build of the library with other
# compile-time settings, as pathtracer_amd/lib/alt_<name>.so (use with PT_LIB_PATH=...); the default build stays.
NAME=$1; shift
cd pathtracer_amd/csrc && mkdir -p build/alt_$NAME
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c pt_kernels.hip -o build/alt_$NAME/pt_kernels.hip.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/alt_$NAME.so build/alt_$NAME/pt_kernels.hip.o build/pt_api.cpp.o build/pt_bvh.cpp.o build/pt_obj.cpp.o build/pt_image.cpp.o -lz
