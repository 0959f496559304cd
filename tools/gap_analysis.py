"""GPU idle-gap analysis of a rocprofv3 --kernel-trace CSV: busy time vs wall span, and the largest gaps
grouped by the kernel that FOLLOWS the gap (the launch the host was late with)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50]) for r in csv.DictReader(open(f))))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2       # analyse the second half (steady state)
rows = rows[skip:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms ({100*(span-busy)/span:.1f} %)")
gaps = collections.defaultdict(lambda: [0, 0])
prev_end = rows[0][1]
for s, e, n in rows[1:]:
    g = max(0, s - prev_end)
    gaps[n][0] += g; gaps[n][1] += 1
    prev_end = max(prev_end, e)
for n, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"  {n:52s} gaps before: {g/1e6:8.3f} ms over {c:5d} launches  ({g/c/1e3:6.1f} us each)")
