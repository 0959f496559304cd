#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python bench.py --no-cpu-baseline > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_b.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "avg", d["roofline"]["avg_launch_ms"])
print("step", {k: v for k, v in d["roofline"]["step"].items() if k != "note"})
print("f32", d.get("f32_mfma_only", {}).get("value")); print("ref", d.get("reference_loop", {}).get("ms_per_step"))
s = d["secondary"]["configs[2]"]
print("secondary", s.get("value"), s.get("ms_per_step"), s.get("hbm_peak_gb"), s.get("error"), s.get("roofline", {}).get("frac"), s.get("roofline", {}).get("mfma_frac"))
for k, v in d["roofline"]["kernels"].items(): print(k, v["launches"], v["ms_total"], v["mfma_frac"])
PY
