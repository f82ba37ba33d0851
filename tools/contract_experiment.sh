#!/bin/bash
# Upper bound of what FMA contraction would buy on the trace kernel (results are NOT bit-exact
# with the oracle in this build; timing only).  Run on the GPU box: tools/contract_experiment.sh "c2 64"
set -e
cd "$(dirname "$0")/.."
args="${1:-c2 64}"
cp renderbaby_amd/librenderbaby_hip.so /tmp/lib_prod.so
for r in 1 2; do echo "[contract off] $(python tools/one_dispatch.py $args)"; done
rm -rf /tmp/csrc_fma && mkdir -p /tmp/csrc_fma/renderbaby_amd /tmp/csrc_fma/include && cp -r renderbaby_amd/csrc /tmp/csrc_fma/renderbaby_amd/ && cp -r include/* /tmp/csrc_fma/include/
sed -i 's/#pragma clang fp contract(off)/#pragma clang fp contract(fast)/' /tmp/csrc_fma/renderbaby_amd/csrc/*.hpp /tmp/csrc_fma/renderbaby_amd/csrc/*.hip
C=/tmp/csrc_fma/renderbaby_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -shared \
  -o renderbaby_amd/librenderbaby_hip.so $C/rb_kernels.hip $C/rb_build.hip $C/rb_runtime.cpp $C/rb_bvh.cpp $C/rb_rccl.cpp -ldl 2>/dev/null
for r in 1 2; do echo "[contract fast] $(python tools/one_dispatch.py $args)"; done
cp /tmp/lib_prod.so renderbaby_amd/librenderbaby_hip.so
