"""rocprofv3 PMC passes of one bench workload (tools/profile_bench.sh) -> profiles/<tag>_<key>_pmc.json:
per-segment constants of the dominant trace kernel.  Units and gfx950 corrections as MI355X_MICROARCH.md
(section HBM) prescribes: FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced read (x2 for such reads: k_accumulate's colour rows; the trace kernels' scattered 16-B loads are not
corrected); WRITE_SIZE is exact for 16-B-per-lane stores.

    python tools/profile_summary.py <out_dir> <tag> <key> "<bench args>"
"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderbaby_amd import _lib

out, tag, key, args = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]


def bench_line(log):
    for line in reversed(open(log).read().splitlines()):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    raise SystemExit(f"no bench line in {log}: " + open(log).read()[-2000:])


def counters(passdir):
    """-> {kernel short name: {counter: (sum over dispatches, n dispatches)}}"""
    agg = {}
    for f in glob.glob(passdir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0]
            d = agg.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, set()])
            d[0] += float(r["Counter_Value"])
            d[1].add(r["Dispatch_Id"])
    return {k: {c: (v[0], len(v[1])) for c, v in d.items()} for k, d in agg.items()}


lines = {p: bench_line(f"{out}/{p}.log") for p in "abcd"}
b = lines["a"]
kernel = b["roofline"]["kernel"].replace("<false>", "")
def is_dom(name):   # the un-instrumented instantiation of the dominant kernel: k_trace<false, ...>, k_trace_bvh<false, false, 256u>, ...
    return name.startswith(kernel + "<false")
steps_run = b["steps"] + b["warmup"]
seg_run = b["config"]["segments_per_step"] * steps_run     # segments the dominant kernel traced in a PMC pass
cnt = {}
for p in "abcd":
    for k, d in counters(f"{out}/{p}").items():
        if is_dom(k):
            cnt.update({c: v[0] for c, v in d.items()})
            cnt.setdefault("_dispatches", d[next(iter(d))][1])
acc = {}
for p in "cd":
    for k, d in counters(f"{out}/{p}").items():
        if k.startswith("k_accumulate"):
            acc.update({c: v[0] for c, v in d.items()})
ms = None
for f in glob.glob(out + "/e/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if is_dom(r["Name"].split("(anonymous namespace)::")[-1].split("(")[0]):
            ms = float(r["AverageNs"]) / 1e6
e = bench_line(f"{out}/e.log")
fetch_b, write_b = cnt.get("FETCH_SIZE", 0.0) * 1024.0, cnt.get("WRITE_SIZE", 0.0) * 1024.0
per_segment = {
    "valu_instr": cnt["SQ_INSTS_VALU"] / seg_run,
    "salu_instr": cnt["SQ_INSTS_SALU"] / seg_run,
    "hbm_bytes": (fetch_b + write_b) / seg_run,
    "hbm_fetch_bytes": fetch_b / seg_run, "hbm_write_bytes": write_b / seg_run,
    "tcp_accesses": cnt.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / seg_run,
    # TCP -> TCC read requests are 64 B each on gfx950 (a 16-B-per-lane wave load = 16 requests of 64 B)
    "l2_bytes": cnt.get("TCP_TCC_READ_REQ_sum", 0.0) * 64.0 / seg_run,
    "vmem_rd_instr": cnt.get("SQ_INSTS_VMEM_RD", 0.0) / seg_run,
    "lds_instr": cnt.get("SQ_INSTS_LDS", 0.0) / seg_run,
}
launches = max(int(cnt["_dispatches"]) // steps_run, 1)
kernel_ms_pmc = None
derived = {
    "valu_lane_utilisation": cnt["SQ_THREAD_CYCLES_VALU"] / (cnt["SQ_ACTIVE_INST_VALU"] * 64.0),
    "wait_any_frac": cnt["SQ_WAIT_ANY"] / cnt["SQ_WAVE_CYCLES"],
    "wait_inst_any_frac": cnt["SQ_WAIT_INST_ANY"] / cnt["SQ_WAVE_CYCLES"],
    "l2_hit_rate": (cnt["TCC_HIT_sum"] / (cnt["TCC_HIT_sum"] + cnt["TCC_MISS_sum"])) if cnt.get("TCC_HIT_sum") else None,
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; per dispatch / 8 / that dispatch's duration (kernel-trace average)
    "clock_GHz": (cnt["GRBM_GUI_ACTIVE"] / cnt["_dispatches"] / 8.0 / (ms * 1e-3) / 1e9) if (ms and cnt.get("GRBM_GUI_ACTIVE")) else None,
    "launches_per_step": launches,
    "kernel_ms_kernel_trace_avg": ms,
    "kernel_ms_bench_hip_events": e["roofline"]["kernel_ms"],
    "Msamples_per_s_unprofiled": e["value"],
}
d = {"command": f"tools/profile_bench.sh {tag} {key} {args}: rocprofv3 --pmc <group> -- python3 bench.py {args} --steps 1 --warmup 1 "
                "--cpu-seconds 0 --no-stats (four counter passes), rocprofv3 --kernel-trace --stats -- python3 bench.py ... --steps 3",
     "workload": b["config"]["workload"], "kernel": kernel, "source_fingerprint": _lib.source_fingerprint(),
     "git_head": os.popen("git -C %s rev-parse --short HEAD 2>/dev/null" % os.path.dirname(os.path.abspath(__file__))).read().strip() or None,
     "segments_in_a_pmc_pass": seg_run, "counters_dominant_kernel": {k: v for k, v in cnt.items()},
     "counters_k_accumulate": acc, "per_segment": per_segment, "derived": derived,
     "hbm_bytes_per_launch": {"trace": (fetch_b + write_b) / max(cnt["_dispatches"], 1),
                              "accumulate_fetch_x2": acc.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0 / max(cnt["_dispatches"], 1),
                              "accumulate_write": acc.get("WRITE_SIZE", 0.0) * 1024.0 / max(cnt["_dispatches"], 1)}}
path = f"profiles/{tag}_{key}_pmc.json"
json.dump(d, open(path, "w"), indent=1)
print(path, json.dumps({"per_segment": per_segment, "derived": derived}, indent=1))
