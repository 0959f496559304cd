"""Turn gpurun_out/prof_TAG (tools/profile_round.sh) into the committed summaries under profiles/:
  TAG_kernel_stats_{sfx}.{csv,md}, TAG_pmc_{sfx}.json (+ pmc_bench_latest.json), TAG_sq_counters_{sfx}.md
usage: python tools/profile_to_md.py TAG "one-line description of the code state" """
import collections, csv, glob, json, os, shutil, subprocess, sys
tag, desc = sys.argv[1], sys.argv[2]
extra = os.environ.get("BENCH_ARGS", "")                 # the BENCH_ARGS the set was collected with, e.g. "--conv bf16"
import re
_m = re.search(r"--batch\s+(\d+)", extra)
batch = int(_m.group(1)) if _m else 32
what = f"B={batch} 1x256x256, " + ("bf16 MFMA conv path (bf16 copies of the conv operands, fp32 master tensors and accumulation)" if "bf16" in extra else "fp32")
sfx = f"bench_b{batch}_256"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
# ---- kernel stats
st = glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(st, f"profiles/{tag}_kernel_stats_{sfx}.csv")
rows = list(csv.DictReader(open(st)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
# steady-state view from the kernel trace: only the last three adam_kernel-delimited steps (the warm-up steps carry
# one-time costs such as the first touch of a freshly grown split-K workspace: single 20 ms wgrad launches)
tr = glob.glob(f"{src}/stats/**/*kernel_trace.csv", recursive=True)
steady = None
if tr:
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(tr[0])))
    adam = [i for i, e in enumerate(ev) if e[2].startswith("adam_kernel")]
    nb = 0 if "--bracket-all" in extra else 3          # bench.py's fully bracketed breakdown loop runs AFTER the timed steps
    if len(adam) >= 4 + nb:
        adam = adam[-(4 + nb):len(adam) - nb]   # the three TIMED optimizer-delimited steps
        ev = ev[adam[0] + 1:adam[-1] + 1]
        agg = collections.defaultdict(lambda: [0, 0])
        for s0, e0, n in ev:
            agg[n][0] += 1
            agg[n][1] += e0 - s0
        steady = (len(adam) - 1, agg)
bench = {}
try:
    bench = json.loads(open(f"{src}/bench_line_under_profiler.json").read())
except Exception:
    pass
with open(f"profiles/{tag}_kernel_stats_{sfx}.md", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline {extra} ({desc})\n\n")
    _w = re.findall(r"--warmup\s+(\d+)", "--warmup 3 " + extra)
    nwarm = int(_w[-1])
    nbk = 0 if "--bracket-all" in extra else 3
    f.write(f"MI355X, {what}; {nwarm + 3 + nbk} training steps traced ({nwarm} warm-up + 3 timed" +
            (f" + {nbk} of bench.py's fully bracketed breakdown loop" if nbk else "") + f").  Total kernel time {tot/1e6:.1f} ms = "
            f"{tot/1e6/(nwarm + 3 + nbk):.1f} ms/step over all of them.")
    if bench:
        dom = bench["roofline"]["kernel"]
        f.write(f"  bench under the profiler: {bench['ms_per_step']} ms/step, {bench['value']} images/s.\n"
                f"bench.py's own HIP-event timing of the dominant kernel in the same run: {dom}, "
                f"{bench['roofline']['launches_timed']} launches, avg {bench['roofline']['avg_launch_ms']} ms "
                f"(this table: the `{dom}` rows).\n")
    if steady:
        nst, agg = steady
        stot = sum(v[1] for v in agg.values())
        f.write(f"\nSteady state (the {nst} timed steps only, from the kernel trace of the same run; the warm-up step carries one-time "
                f"costs): {stot/1e6/nst:.1f} ms of kernel time per step.\n")
        f.write("\n| kernel (steady state) | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            f.write(f"| `{n[:90]}` | {c} | {t/1e6:.2f} | {t/c/1e3:.1f} | {100*t/stot:.2f} |\n")
    f.write("\nAll traced launches (rocprofv3's own `--stats` table, warm-up step included):\n")
    f.write("\n| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:40]:
        f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |\n")
# ---- PMC traffic
out = f"profiles/{tag}_pmc_{sfx}.json"
subprocess.run([sys.executable, "tools/pmc_parse.py", f"{src}/fetch", f"{src}/write", "--json", out], check=True, stdout=subprocess.DEVNULL)
if not extra:
    shutil.copy(out, "profiles/pmc_bench_latest.json")       # the default (fp32) run is what bench.py's `traffic` quotes
else:                                                        # other configurations: looked up by conv path and batch
    shutil.copy(out, f"profiles/pmc_bench_{'bf16' if 'bf16' in extra else 'f32'}_b{batch}_latest.json")
# ---- SQ counters
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for fcsv in glob.glob(f"{src}/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fcsv)):
        vals[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for fcsv in glob.glob(f"{src}/sq/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fcsv)):
        dur[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(f"profiles/{tag}_sq_counters_{sfx}.md", "w") as f:
    f.write(f"# rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -- python bench.py --steps 3 --warmup 3 {extra} ({desc})\n\n")
    f.write("effective clock = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs).\n"
            "Profiled passes run slower than un-profiled ones (same guide), so durations here are longer than in the kernel-stats table.\n\n")
    f.write("Wave time split (fractions of SQ_WAVE_CYCLES): issuing = SQ_ACTIVE_INST_ANY, waiting to issue = SQ_WAIT_INST_ANY (matrix pipe busy, dependencies), "
            "parked = SQ_WAIT_ANY (s_waitcnt, barrier).\n\n")
    f.write("| kernel | launches | total ms | effective clock GHz | MFMA busy | issuing | waiting to issue | parked | VALU per MFMA |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|\n")
    for k in sorted(dur, key=lambda k: -sum(dur[k]))[:16]:
        g = vals[k].get("GRBM_GUI_ACTIVE", [])
        m = vals[k].get("SQ_VALU_MFMA_BUSY_CYCLES", [])
        if not g:
            continue
        cyc = sum(g) / 8
        wc = sum(vals[k].get("SQ_WAVE_CYCLES", [])) or 1.0
        fr = lambda c: sum(vals[k].get(c, [])) / wc
        nm = sum(vals[k].get("SQ_INSTS_MFMA", []))
        vpm = f"{(sum(vals[k].get('SQ_INSTS_VALU', [])) - nm) / nm:.1f}" if nm else "-"
        f.write(f"| `{k}` | {len(dur[k])} | {sum(dur[k])/1e6:.2f} | {cyc / sum(dur[k]):.2f} | {sum(m) / 1024 / cyc:.3f} | "
                f"{fr('SQ_ACTIVE_INST_ANY'):.2f} | {fr('SQ_WAIT_INST_ANY'):.2f} | {fr('SQ_WAIT_ANY'):.2f} | {vpm} |\n")
print("written profiles/ for", tag)
