#!/bin/bash
# Pass occupancy of k_trace_chunk: a copy of the kernel sources gets the counting blocks of tools/ablate/rb_profile.patch
# (RB_CHUNK_PROFILE=1: node passes and the lanes in them, leaf rounds and the pairs in them, finish passes; =2: outer iterations,
# lanes shaded, leaf phases, walking lanes), is built into renderbaby_amd/variants/lib_prof<n>.so and run by
# `tools/chunk_probe.py prof`.  The product sources carry no counting code.
set -e
cd "$(dirname "$0")/.."
R="$(pwd)"; W=/tmp/rb_prof_src; rm -rf $W; mkdir -p $W/renderbaby_amd $R/renderbaby_amd/variants
cp -r renderbaby_amd/csrc $W/renderbaby_amd/csrc; cp -r include $W/include
(cd $W && patch -p0 -s < $R/tools/ablate/rb_profile.patch)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize"
C=$W/renderbaby_amd/csrc
for n in 1 2; do
  /opt/rocm/bin/hipcc $FLAGS -DRB_CHUNK_PROFILE=$n -shared -o renderbaby_amd/variants/lib_prof$n.so $C/rb_kernels.hip $C/rb_build.hip $C/rb_runtime.cpp $C/rb_bvh.cpp $C/rb_rccl.cpp -ldl
  RB_LIBRARY_PATH=renderbaby_amd/variants/lib_prof$n.so python tools/chunk_probe.py prof "$@"
done
