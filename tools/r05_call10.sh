#!/bin/bash
# round 5, GPU call 10: after the FLAT -> global typed accesses (k_sao_rgb's RGB stores, the batched colour kernels, k_mc): GPU tier, main workload, Main10 workload
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
timeout 600 python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/c10_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c10_tests.log | cut -c1-200
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print("%-12s" % sys.argv[1], d.get("value_resident", d["value"]), "ms/step", d["ms_per_step"], {k: round(v["avg_us"] / 1e3, 2) for k, v in d["kernels"].items()})
except Exception as e: print(sys.argv[1], "no line", e)
PY
}
for rep in 1 2; do
  timeout 200 python bench.py --only-main --steps 4 --warmup 1 > gpurun_out/c10_main_$rep.json 2> gpurun_out/c10_main_$rep.err; show main/$rep gpurun_out/c10_main_$rep.json
done
timeout 200 python bench.py --workload main10_4k --only-main --steps 4 --warmup 1 > gpurun_out/c10_main10.json 2> gpurun_out/c10_main10.err; show main10 gpurun_out/c10_main10.json
timeout 200 python bench.py --workload still1080 --only-main --steps 4 --warmup 1 > gpurun_out/c10_1080.json 2> gpurun_out/c10_1080.err; show 1080p gpurun_out/c10_1080.json
