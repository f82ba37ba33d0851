#!/bin/bash
# colour-store write combining (ColorRing) on / off: k_trace time and HBM write traffic on C2 (64 spp), on the GPU box.
# Builds a second library with -DRB_COLOR_COMBINE=0 into /tmp and selects it with RB_LIBRARY_PATH.
cd "$(dirname "$0")/.."
R=$(pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -DRB_COLOR_COMBINE=0 -shared -o /tmp/lib_comb0.so renderbaby_amd/csrc/rb_kernels.hip renderbaby_amd/csrc/rb_build.hip renderbaby_amd/csrc/rb_runtime.cpp renderbaby_amd/csrc/rb_bvh.cpp renderbaby_amd/csrc/rb_rccl.cpp -ldl 2>/dev/null || exit 1
for v in "" /tmp/lib_comb0.so; do
  export RB_LIBRARY_PATH=$v
  tag=$([ -z "$v" ] && echo ring || echo direct)
  for r in 1 2; do echo "[$tag] $(python tools/one_dispatch.py c2 64 0 3 2>/dev/null | tail -1)"; done
  (cd /tmp && TMPDIR=/tmp rocprofv3 --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/comb_w_$tag -- python3 $R/tools/one_dispatch.py c2 64 0 1 > /dev/null 2>&1)
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob("$R/gpurun_out/comb_w_$tag/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "WRITE_SIZE" and "k_trace" in row["Kernel_Name"]: acc["k_trace"] += float(row["Counter_Value"])
for k, v in acc.items(): print("[$tag] WRITE_SIZE", k, "%.3f GB" % (v * 1024 / 1e9), "(payload 2.123 GB: 2 073 600 pixels x 64 samples x 16 B)")
PY
done
