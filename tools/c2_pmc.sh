#!/bin/bash
# VALU picture of the C2 trace kernel (what bench.py quotes under roofline.valu):
#   tools/c2_pmc.sh r01   ->  profiles/r01_c2_64spp_final_pmc.json
set -e
tag="${1:-r01}"
R="$(cd "$(dirname "$0")/.." && pwd)"
out="$R/gpurun_out/c2_pmc_$tag"; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d "$out/a" -- python3 "$R/tools/one_dispatch.py" c2 64 3 2 > /dev/null 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_SALU GRBM_GUI_ACTIVE -d "$out/b" -- python3 "$R/tools/one_dispatch.py" c2 64 3 2 > /dev/null 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/c" -- python3 "$R/tools/one_dispatch.py" c2 64 3 2 > /dev/null 2>&1
cd "$R"
python - "$out" "$tag" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
agg, n = {}, {}
for f in glob.glob(out + "/[ab]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace<false>" not in r["Kernel_Name"]:
            continue
        c = r["Counter_Name"]
        agg[c] = agg.get(c, 0.0) + float(r["Counter_Value"])
        n.setdefault(c, set()).add(r["Dispatch_Id"])
cnt = {c: agg[c] / len(n[c]) for c in agg}          # per dispatch
ms = None
for f in glob.glob(out + "/c/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace<false>" in r["Name"]:
            ms = float(r["AverageNs"]) / 1e6
segments = 684142608                                  # C2 at 64 spp (device counter, equal to the oracle's)
wave_iters = segments / 64.0
d = {"command": "tools/c2_pmc.sh: rocprofv3 --pmc ... -- python3 tools/one_dispatch.py c2 64 3 2 (C2 Cornell 1920x1080, 64 spp, depth 8; k_trace<false> per-dispatch averages)",
     "kernel_ms": ms, "counters": cnt,
     "derived": {"valu_instr_per_wave_iteration": cnt["SQ_INSTS_VALU"] / wave_iters,
                 "salu_instr_per_wave_iteration": cnt["SQ_INSTS_SALU"] / wave_iters,
                 "valu_lane_utilisation": cnt["SQ_THREAD_CYCLES_VALU"] / (cnt["SQ_ACTIVE_INST_VALU"] * 64.0),
                 "clock_GHz": cnt["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e-3) / 1e9 if ms else None,
                 "wait_any_frac": cnt["SQ_WAIT_ANY"] / cnt["SQ_WAVE_CYCLES"],
                 "wait_inst_any_frac": cnt["SQ_WAIT_INST_ANY"] / cnt["SQ_WAVE_CYCLES"]}}
json.dump(d, open(f"profiles/{tag}_c2_64spp_final_pmc.json", "w"), indent=1)
print(json.dumps(d["derived"], indent=1), "kernel_ms", ms)
PY
mkdir -p "$R/gpurun_out/profiles_$tag" && cp "profiles/${tag}_c2_64spp_final_pmc.json" "$R/gpurun_out/profiles_$tag/"
