#!/bin/bash
# Cost attribution for k_trace on the GPU box: rebuild with RB_ABLATE=n (stage n executed twice)
# and time C2-short.  0 = production, 1 = BVH/triangles, 2 = spheres, 3 = RNG unit vector, 4 = start_path.
set -e
cd "$(dirname "$0")/.."
cp renderbaby_amd/librenderbaby_hip.so /tmp/lib_prod.so
for n in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -DRB_ABLATE=$n -shared -o renderbaby_amd/librenderbaby_hip.so renderbaby_amd/csrc/rb_kernels.hip renderbaby_amd/csrc/rb_build.hip renderbaby_amd/csrc/rb_runtime.cpp renderbaby_amd/csrc/rb_bvh.cpp renderbaby_amd/csrc/rb_rccl.cpp -ldl 2>/dev/null
  echo "ABLATE=$n $(python tools/one_dispatch.py c2 64 3 3)"
done
cp /tmp/lib_prod.so renderbaby_amd/librenderbaby_hip.so
