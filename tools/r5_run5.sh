#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -m gpu -q -k "bf16_conv_path or presplit_storage_step or 8x8_map or partially_frozen or config3" > gpurun_out/gputest_d.log 2>&1
tail -8 gpurun_out/gputest_d.log
timeout -k 10 900 python bench.py --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_a.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "avg", d["roofline"]["avg_launch_ms"])
print("step", d["roofline"]["step"])
print("f32", d.get("f32_mfma_only")); print("ref", d.get("reference_loop"))
s = d["secondary"]["configs[2]"]
print("secondary", s.get("value"), s.get("ms_per_step"), s.get("hbm_peak_gb"), s.get("error"), s.get("roofline", {}).get("frac"), s.get("roofline", {}).get("mfma_frac"))
for k, v in d["roofline"]["kernels"].items(): print(k, v["launches"], v["ms_total"], v["mfma_frac"])
PY
