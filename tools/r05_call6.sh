#!/bin/bash
# round 5, GPU call 6: the whole GPU tier with per-test durations (torch imported once before it: a cold import takes minutes on a fresh box), then the default bench line
mkdir -p gpurun_out
python -c "import torch; print('torch', torch.__version__, torch.cuda.is_available())" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q --timeout 400 --durations=15 > gpurun_out/c6_tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/c6_tests.log | cut -c1-200
timeout 700 python bench.py > gpurun_out/c6_bench.json 2> gpurun_out/c6_bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/c6_bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/c6_bench.json"))
print("value", d["value"], "resident", d.get("value_resident"), "ms", d["ms_per_step"], "verified", d.get("verified", {}).get("planes_match"), d.get("verified", {}).get("rgb_match"))
print({k: round(v["avg_us"] / 1e3, 2) for k, v in d["kernels"].items()})
print("roofline", d["roofline"]["frac"], d["roofline"].get("limited_by"), "issue", d["issue_roofline"]["kernels"]["parse"]["frac_of_scalar_issue_peak"], d["issue_roofline"].get("clock_ghz"))
print("single", d.get("single_still"), "\nseq", json.dumps(d.get("sequence_tracks"))[:900])
print("configs", {k: (v.get("mpixel_s"), v.get("ms")) for k, v in d.get("baseline_configs", {}).items()})
print("dropin", json.dumps(d.get("dropin_through_libheif"))[:600])
print("cpu", d.get("cpu_baseline", {}).get("value"), "lib", d.get("native_library"))
PY
