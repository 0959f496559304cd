#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python bench.py > gpurun_out/bench_c.json 2> gpurun_out/bench_c.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_c.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "avg", d["roofline"]["avg_launch_ms"], "traffic", d["roofline"].get("traffic"))
print("step", {k: v for k, v in d["roofline"]["step"].items() if k != "note"})
print("f32", d.get("f32_mfma_only", {}).get("value")); print("ref", d.get("reference_loop", {}).get("ms_per_step"))
print("cpu", d.get("cpu_baseline"))
s = d["secondary"]["configs[2]"]
print("secondary", s.get("value"), s.get("ms_per_step"), s.get("hbm_peak_gb"), s.get("error"), s.get("roofline", {}).get("frac"), s.get("roofline", {}).get("mfma_frac"))
for k, v in d["roofline"]["kernels"].items(): print(k, v["launches"], v["ms_total"], v["mfma_frac"])
print("streaming", d["roofline"]["streaming"]["achieved"], d["roofline"]["streaming"]["kernel"])
PY
