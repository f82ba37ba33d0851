#!/bin/bash
# colour-store variants: time and HBM write traffic of k_trace on C2 (64 spp)
cd "$(dirname "$0")/.."
R=$(pwd)
cp renderbaby_amd/librenderbaby_hip.so /tmp/lib_prod.so
for nt in 0 1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -DRB_COLOR_STORE_NT=$nt -shared -o renderbaby_amd/librenderbaby_hip.so renderbaby_amd/csrc/rb_kernels.hip renderbaby_amd/csrc/rb_build.hip renderbaby_amd/csrc/rb_runtime.cpp renderbaby_amd/csrc/rb_bvh.cpp 2>/dev/null
  echo "[nt=$nt] $(python tools/one_dispatch.py c2 64 0 3)"
  (cd /tmp && TMPDIR=/tmp rocprofv3 --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/nt_w$nt -- python3 $R/tools/one_dispatch.py c2 64 0 1 > /dev/null 2>&1)
  (cd /tmp && TMPDIR=/tmp rocprofv3 --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/nt_f$nt -- python3 $R/tools/one_dispatch.py c2 64 0 1 > /dev/null 2>&1)
  python tools/pmc_summary.py gpurun_out/nt_w$nt gpurun_out/nt_f$nt | grep -A3 "k_trace\|k_accum" | grep "k_trace\|k_accum\|SIZE"
done
cp /tmp/lib_prod.so renderbaby_amd/librenderbaby_hip.so
