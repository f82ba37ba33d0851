#!/bin/bash
# A profiling build of the library: a copy of the sources gets the counting code of tools/ablate/rb_profile.patch (the product
# sources carry none) and is built into renderbaby_amd/variants/lib_walkprof.so, in the dev container (hipcc cross-compiles),
# so that it travels to the GPU box.  There:
#   RB_LIBRARY_PATH=$PWD/renderbaby_amd/variants/lib_walkprof.so python tools/sph_profile.py c4 8        k_trace_sph's passes
#   RB_LIBRARY_PATH=$PWD/renderbaby_amd/variants/lib_walkprof.so python tools/ktrace_phases.py c2 64     k_trace's lanes per phase
set -e
cd "$(dirname "$0")/.."
R="$(pwd)"; W=/tmp/rb_walkprof_src; rm -rf $W; mkdir -p $W/renderbaby_amd $R/renderbaby_amd/variants
cp -r renderbaby_amd/csrc $W/renderbaby_amd/csrc; cp -r include $W/include
(cd $W && patch -p0 -s < $R/tools/ablate/rb_profile.patch)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize"
C=$W/renderbaby_amd/csrc
/opt/rocm/bin/hipcc $FLAGS "$@" -shared -o renderbaby_amd/variants/lib_walkprof.so $C/rb_kernels.hip $C/rb_build.hip $C/rb_runtime.cpp $C/rb_bvh.cpp $C/rb_rccl.cpp -ldl
echo renderbaby_amd/variants/lib_walkprof.so
