#!/bin/bash
# colour-store write combining (ColorRing) on/off: k_trace time and HBM write traffic on C2 (64 spp).
# Libraries: the in-tree one and renderbaby_amd/variants/lib_comb0.so (-DRB_COLOR_COMBINE=0), built beforehand.
cd "$(dirname "$0")/.."
R=$(pwd)
for v in "" renderbaby_amd/variants/lib_comb0.so; do
  export RB_LIBRARY_PATH=${v:+$R/$v}
  tag=$([ -z "$v" ] && echo comb1 || echo comb0)
  for r in 1 2; do echo "[$tag] $(python tools/one_dispatch.py c2 64 0 3)"; done
  (cd /tmp && TMPDIR=/tmp rocprofv3 --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/comb_w_$tag -- python3 $R/tools/one_dispatch.py c2 64 0 1 > /dev/null 2>&1)
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob("$R/gpurun_out/comb_w_$tag/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "WRITE_SIZE": acc[row["Kernel_Name"].split("(")[0][:60]] += float(row["Counter_Value"])
for k, v in acc.items(): print("[$tag] WRITE_SIZE", k, "%.3f GB" % (v * 1024 / 1e9))
PY
done
