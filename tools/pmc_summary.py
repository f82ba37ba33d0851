"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py <dir> [<dir> ...]"""
import csv, glob, json, sys
from collections import defaultdict
out = defaultdict(lambda: defaultdict(float))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0]
            out[k][r["Counter_Name"]] += float(r["Counter_Value"])
print(json.dumps(out, indent=1))
