"""Parse rocprofv3 --pmc CSV output: per-kernel counter averages.  With --json OUT writes the
calibrated HBM bytes per launch: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE counts
half the bytes on gfx950 (MI355X_MICROARCH.md §HBM; re-calibrated here on 512 MiB copies at 4 and
16 B/lane: FETCH_SIZE = 262 164 KB for 524 288 KB read, WRITE_SIZE = 524 288 KB exact)."""
import collections
import csv
import glob
import json
import sys

args = sys.argv[1:]
out = None
if "--json" in args:
    i = args.index("--json")
    out = args[i + 1]
    del args[i:i + 2]
vals = collections.defaultdict(list)
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:64]
            vals[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
res = collections.defaultdict(dict)
for (k, c), v in sorted(vals.items()):
    print(f"{k:66s} {c:12s} n={len(v):4d} avg={sum(v)/len(v):14.1f} KB")
    res[k][c] = {"launches": len(v), "avg_KB": sum(v) / len(v)}
if out:
    for k, d in res.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"]["avg_KB"] + d["WRITE_SIZE"]["avg_KB"]) * 1024
    json.dump(res, open(out, "w"), indent=1)
