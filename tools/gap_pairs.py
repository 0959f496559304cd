"""Which kernel PRECEDES the large idle gaps of a rocprofv3 --kernel-trace CSV (steady-state half)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e6
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in csv.DictReader(open(f))))
rows = rows[len(rows) // 2:]
pairs = collections.defaultdict(lambda: [0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g > thr:
        pairs[(n0, n1)][0] += g; pairs[(n0, n1)][1] += 1
for (a, b), (g, c) in sorted(pairs.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"{a:44s} -> {b:44s} {g/1e6:8.2f} ms in {c:3d} gaps")
