#!/bin/bash
# Cost attribution for k_trace on the GPU box: a copy of the kernel sources gets the duplicate-a-stage blocks of
# tools/ablate/rb_ablate.patch (RB_ABLATE=n executes stage n twice on perturbed-but-equal inputs and folds the result into
# nothing observable, so time[n] - time[0] is that stage's cost), is built into renderbaby_amd/variants/ and timed on
# C2-short.  0 = production, 1 = BVH/triangles, 2 = spheres, 3 = RNG unit vector, 4 = start_path.
# The product sources carry no ablation code.
set -e
cd "$(dirname "$0")/.."
R="$(pwd)"; W=/tmp/rb_ablate_src; rm -rf $W; mkdir -p $W/renderbaby_amd $R/renderbaby_amd/variants
cp -r renderbaby_amd/csrc $W/renderbaby_amd/csrc; cp -r include $W/include
(cd $W && patch -p0 -s < $R/tools/ablate/rb_ablate.patch)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize"
C=$W/renderbaby_amd/csrc
for n in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc $FLAGS -DRB_ABLATE=$n -shared -o renderbaby_amd/variants/lib_ablate$n.so $C/rb_kernels.hip $C/rb_build.hip $C/rb_runtime.cpp $C/rb_bvh.cpp $C/rb_rccl.cpp -ldl 2>/dev/null
  echo "ABLATE=$n $(RB_LIBRARY_PATH=renderbaby_amd/variants/lib_ablate$n.so python tools/one_dispatch.py c2 64 3 3)"
done
