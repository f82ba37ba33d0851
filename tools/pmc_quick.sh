#!/bin/bash
# Quick counters of one dispatch (not the bench's evidence: tools/profile_bench.sh is): two rocprofv3 --pmc passes over
#   python3 tools/one_dispatch.py <args>     and the per-segment figures of the kernel named by $KERNEL (default k_trace).
#   KERNEL=k_trace_sph RB_SPH_TREE=host tools/pmc_quick.sh c4 8 0 1
R="$(cd "$(dirname "$0")/.." && pwd)"
out="$R/gpurun_out/pmcq_$$"; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU -d "$out/a" -- python3 "$R/tools/one_dispatch.py" "$@" > "$out/a.log" 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d "$out/b" -- python3 "$R/tools/one_dispatch.py" "$@" > "$out/b.log" 2>&1
cd "$R"
python3 - "$out" "${KERNEL:-k_trace}" <<'PY'
import csv, glob, re, sys
out, kern = sys.argv[1], sys.argv[2]
seg = None; line = ""
for l in open(out + "/a.log"):
    m = re.search(r"segments (\d+) Mseg/s", l)
    if m: seg = int(m.group(1)); line = l.strip()
cnt = {}; nd = {}
for p in "ab":
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0]
            if k.startswith(kern + "<"):
                cnt[r["Counter_Name"]] = cnt.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                nd.setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
n = len(next(iter(nd.values()))) if nd else 0
print(line)
if not cnt or not seg: raise SystemExit("no counters for " + kern + "; " + open(out + "/a.log").read()[-800:])
seg *= n   # every dispatch of the process traced the same frame
print(f"{kern}: {n} dispatches; per segment: VALU {cnt['SQ_INSTS_VALU']/seg:.1f}  SALU {cnt['SQ_INSTS_SALU']/seg:.1f}  L1 accesses {cnt['TCP_TOTAL_CACHE_ACCESSES_sum']/seg:.1f}  "
      f"vmem rd {cnt['SQ_INSTS_VMEM_RD']/seg:.2f}  lds {cnt['SQ_INSTS_LDS']/seg:.2f}  L2 bytes {cnt['TCP_TCC_READ_REQ_sum']*64/seg:.0f}")
print(f"lane utilisation {cnt['SQ_THREAD_CYCLES_VALU']/(cnt['SQ_ACTIVE_INST_VALU']*64):.3f}  wait_any {cnt['SQ_WAIT_ANY']/cnt['SQ_WAVE_CYCLES']:.3f}  wait_inst_any {cnt['SQ_WAIT_INST_ANY']/cnt['SQ_WAVE_CYCLES']:.3f}  "
      f"L2 hit {cnt['TCC_HIT_sum']/(cnt['TCC_HIT_sum']+cnt['TCC_MISS_sum']):.3f}  LDS bank-conflict cycles/segment {cnt.get('SQ_LDS_BANK_CONFLICT',0)/seg:.1f}")
PY
