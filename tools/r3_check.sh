#!/bin/bash
# development GPU run (round 3): GPU test tier (stop at the first failure), the main bench line, optionally the parser's PMC passes
# usage: bash tools/r3_check.sh <tag> [pmc]
tag=$1
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/${tag}_tests.log
timeout 300 python bench.py --only-main --steps 3 --warmup 1 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${tag}_bench.json"))
    print("value", d["value"], "ms", d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
except Exception as e:
    print("no bench line", e); print(open("gpurun_out/${tag}_bench.err").read()[-1500:])
PY
if [ "$2" = "pmc" ]; then
  bash tools/prof_parse_pmc.sh ${tag} --batch 512 > gpurun_out/${tag}_pmc.txt 2>&1
  cat gpurun_out/${tag}_pmc.txt
fi
