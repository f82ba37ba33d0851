"""Turn rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate runs) of a bench.py run into
profiles/traffic_<tag>.json, applying the gfx950 corrections of MI355X_MICROARCH.md (section HBM):
FETCH_SIZE/WRITE_SIZE are in KB; FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
streaming read (k_accumulate's 16 B/lane reads), WRITE_SIZE is exact for 16-B-per-lane stores.

  python tools/make_traffic.py <fetch_csv> <write_csv> <workload> <launches_per_step> <out_json>
"""
import csv, collections, json, sys

fetch_csv, write_csv, workload, out = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]


def per_kernel(path, counter):
    agg = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        key = "k_trace" if "k_trace" in name else ("k_accumulate" if "k_accumulate" in name else None)
        if "<true>" in name:
            key = None  # the instrumented (stats) pass is not the timed kernel
        if key:
            agg[key] += float(r["Counter_Value"])
            n[key].add(r["Dispatch_Id"])
    return {k: (v / len(n[k]), len(n[k])) for k, v in agg.items()}


f = per_kernel(fetch_csv, "FETCH_SIZE")
w = per_kernel(write_csv, "WRITE_SIZE")
res = {"workload": workload, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py`; "
       "KB -> bytes x1024; FETCH_SIZE x2 (gfx950 reports half of wide streaming reads)", "per_kernel_launch": {}}
total = 0.0
for k in sorted(set(f) | set(w)):
    fb = f.get(k, (0, 0))[0] * 1024 * 2
    wb = w.get(k, (0, 0))[0] * 1024
    res["per_kernel_launch"][k] = {"fetch_bytes": fb, "write_bytes": wb, "launches_averaged": f.get(k, (0, 0))[1]}
    total += fb + wb
# one bench step = (trace + accumulate) x number of launch chunks; report the dominant kernel's launch
res["hbm_bytes_per_launch"] = res["per_kernel_launch"].get("k_trace", {}).get("fetch_bytes", 0) + res["per_kernel_launch"].get("k_trace", {}).get("write_bytes", 0)
res["hbm_bytes_per_trace_plus_accumulate_pair"] = total
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
