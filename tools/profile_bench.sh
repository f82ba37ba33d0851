#!/bin/bash
# rocprofv3 evidence for one bench workload, on the GPU box:
#   tools/profile_bench.sh <tag> <key> [bench.py args ...]       e.g.  tools/profile_bench.sh r02 c2
#                                                                       tools/profile_bench.sh r02 c3_reference --workload c3 --walk reference
# -> profiles/<tag>_<key>_pmc.json          per-segment constants of the dominant (trace) kernel, stamped with the
#                                           source fingerprint of the build (bench.py's roofline reads it)
#    profiles/<tag>_<key>_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the same command
#    profiles/<tag>_<key>_bench.json        the bench line itself (run last, so it sees the fresh profile)
# PMC passes are separate runs with counters only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one
# pass; gpurun refuses --pmc together with trace domains).
set -e
tag="$1"; key="$2"; shift 2
R="$(cd "$(dirname "$0")/.." && pwd)"
out="$R/gpurun_out/prof_${tag}_${key}"; rm -rf "$out"; mkdir -p "$out"
args=("$@"); [ ${#args[@]} -eq 0 ] && args=(--workload "$key")
short=("${args[@]}" --steps 1 --warmup 1 --cpu-seconds 0 --no-stats --no-end-to-end)   # counters are summed over every dispatch of the process
cd /tmp && export TMPDIR=/tmp
echo "[profile $key] PMC pass A (SQ)"
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU -d "$out/a" -- python3 "$R/bench.py" "${short[@]}" > "$out/a.log" 2>&1
echo "[profile $key] PMC pass B (clock, L1, L2)"
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d "$out/b" -- python3 "$R/bench.py" "${short[@]}" > "$out/b.log" 2>&1
echo "[profile $key] PMC pass C (FETCH_SIZE)"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$out/c" -- python3 "$R/bench.py" "${short[@]}" > "$out/c.log" 2>&1
echo "[profile $key] PMC pass D (WRITE_SIZE)"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$out/d" -- python3 "$R/bench.py" "${short[@]}" > "$out/d.log" 2>&1
echo "[profile $key] kernel trace"
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/e" -- python3 "$R/bench.py" "${args[@]}" --steps 3 --warmup 1 --cpu-seconds 0 --no-end-to-end > "$out/e.log" 2>&1
cd "$R"
python3 tools/profile_summary.py "$out" "$tag" "$key" "${args[*]}"
cp "$(ls $out/e/*/*kernel_stats.csv | head -1)" "profiles/${tag}_${key}_kernel_stats.csv"
echo "[profile $key] bench line"
python3 bench.py "${args[@]}" > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
tail -1 "$out/bench.json" > "profiles/${tag}_${key}_bench.json"
mkdir -p "$R/gpurun_out/profiles_$tag" && cp profiles/${tag}_${key}_* "$R/gpurun_out/profiles_$tag/"
python3 - "profiles/${tag}_${key}_bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("[bench]", d["config"]["workload"][:40], "value %.1f Msamples/s" % d["value"], "kernel", r["kernel"], "bound", r["bound"],
      "frac", r["frac"], "hbm", (r.get("hbm") or {}).get("frac"), "l2", (r.get("l2") or {}).get("frac"), r.get("note", ""))
PY
