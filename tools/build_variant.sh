#!/bin/bash
# A variant of the library with compile-time knobs, built here (hipcc cross-compiles) so that it travels to the GPU box:
#   tools/build_variant.sh <name> -DRB_FOO=1 ...   ->  renderbaby_amd/variants/lib_<name>.so   (use with RB_LIBRARY_PATH)
# The two device translation units and the two host files that share their knobs are rebuilt; the rest comes from the last `make`.
set -e
cd "$(dirname "$0")/.."
name="$1"; shift
mkdir -p renderbaby_amd/variants build/obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o build/obj/rb_kernels.$name.o renderbaby_amd/csrc/rb_kernels.hip
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o build/obj/rb_build.$name.o renderbaby_amd/csrc/rb_build.hip        # (RB_SPH_LEAF ...)
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o build/obj/rb_bvh.$name.o renderbaby_amd/csrc/rb_bvh.cpp          # (host-side knobs: RB_CHUNK_TRIS ...)
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o build/obj/rb_runtime.$name.o renderbaby_amd/csrc/rb_runtime.cpp
/opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o renderbaby_amd/variants/lib_$name.so build/obj/rb_kernels.$name.o \
    build/obj/rb_build.$name.o build/obj/rb_runtime.$name.o build/obj/rb_bvh.$name.o build/obj/rb_rccl.cpp.o -ldl
echo "renderbaby_amd/variants/lib_$name.so"
