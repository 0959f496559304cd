"""Per-kernel averages of every counter collected by tools/pmc_diag.sh, grouped by (kernel, grid size)."""
import collections, csv, glob, sys
vals = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:48]
        key = (name, r.get("Grid_Size", "?"))
        vals[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
per = collections.defaultdict(dict)
for (key, c), v in vals.items():
    v = v[2:] if len(v) > 2 else v          # drop the warm-up launches
    per[key][c] = sum(v) / len(v)
for key in sorted(per):
    d = per[key]
    if "SQ_WAVE_CYCLES" not in d or d["SQ_WAVE_CYCLES"] < 1e6:
        continue
    wc = d["SQ_WAVE_CYCLES"]
    print(f"\n== {key[0]} grid {key[1]}")
    for c in sorted(d):
        extra = f"  ({d[c] / wc:6.3f} of WAVE_CYCLES)" if c.startswith("SQ_WAIT") or c.startswith("SQ_ACTIVE") else ""
        print(f"   {c:32s} {d[c]:16.1f}{extra}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        print(f"   MFMA busy = {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
    if d.get("SQ_INSTS_MFMA", 0) > 0:
        m = d["SQ_INSTS_MFMA"]
        print("   per MFMA: VALU %.2f  LDS %.2f  VMEM_RD %.3f  SALU %.2f  SMEM %.3f" % (
            (d.get("SQ_INSTS_VALU", 0) - m) / m, d.get("SQ_INSTS_LDS", 0) / m, d.get("SQ_INSTS_VMEM_RD", 0) / m,
            d.get("SQ_INSTS_SALU", 0) / m, d.get("SQ_INSTS_SMEM", 0) / m))
