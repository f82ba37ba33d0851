#!/bin/bash
# A/B builds on the GPU box: tools/ab.sh "<one_dispatch args>" -DFOO=0 -DFOO=1 ...
set -e
cd "$(dirname "$0")/.."
args="$1"; shift
cp renderbaby_amd/librenderbaby_hip.so /tmp/lib_prod.so
for def in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math $def -shared -o renderbaby_amd/librenderbaby_hip.so renderbaby_amd/csrc/rb_kernels.hip renderbaby_amd/csrc/rb_build.hip renderbaby_amd/csrc/rb_runtime.cpp renderbaby_amd/csrc/rb_bvh.cpp renderbaby_amd/csrc/rb_rccl.cpp -ldl 2>/dev/null
  for r in 1 2; do echo "[$def] $(python tools/one_dispatch.py $args)"; done
done
cp /tmp/lib_prod.so renderbaby_amd/librenderbaby_hip.so
