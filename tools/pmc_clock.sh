#!/bin/bash
# Clock and matrix-pipe occupancy per kernel for a library variant: one --pmc pass (GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES,
# SQ_BUSY_CYCLES, SQ_WAIT_ANY, SQ_WAVE_CYCLES) + kernel trace over tools/pmc_kernels.py.   tools/pmc_clock.sh TAG   (ONET_HIP_LIB selects)
set -e
TAG=${1:-clk}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/p -o c --output-format csv -- python3 $R/tools/pmc_kernels.py > $OUT/p.log 2>&1 || echo "pass failed"
python3 - $OUT <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
dur = {}
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r["Kernel_Name"])
cnt = collections.defaultdict(dict)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        cnt[r["Dispatch_Id"]]["_grid"] = r.get("Grid_Size", "?")
agg = collections.defaultdict(list)
for d, c in cnt.items():
    if d not in dur or "GRBM_GUI_ACTIVE" not in c or c.get("SQ_WAVE_CYCLES", 0) < 1e6:
        continue
    ns, name = dur[d]
    name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:52]
    agg[(name, c["_grid"], round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1e7))].append((ns, c))
print(f"{'kernel':54s} {'us':>8s} {'GHz':>6s} {'MFMA busy':>9s} {'parked':>7s} {'issue-stall':>11s}")
for k in sorted(agg):
    v = agg[k][1:] if len(agg[k]) > 2 else agg[k]
    ns = sum(x[0] for x in v) / len(v)
    g = lambda n: sum(x[1].get(n, 0) for x in v) / len(v)
    gui = g("GRBM_GUI_ACTIVE") / 8
    print(f"{k[0]:54s} {ns/1e3:8.1f} {gui/ns:6.3f} {g('SQ_VALU_MFMA_BUSY_CYCLES')/(gui*1024):9.3f} {g('SQ_WAIT_ANY')/g('SQ_WAVE_CYCLES'):7.3f} {g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES'):11.3f}")
PY
