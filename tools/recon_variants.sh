#!/bin/bash
# inside gpurun: the 8-bit reconstruction kernel's development variants (HIPDEC_RECON_VARIANT, recon_kernel.hip:launch_recon): kernel times on
# 1024 4K stills, then the decode parity tests under every non-baseline variant side by side
mkdir -p gpurun_out
for v in ${VARIANTS:-3 0 1 2}; do
  HIPDEC_RECON_VARIANT=$v timeout 120 python bench.py --only-main --steps 2 --warmup 1 --batch 1024 --distinct 64 > gpurun_out/reconv_$v.json 2> gpurun_out/reconv_$v.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/reconv_$v.json")); print("variant $v", d["value"], {k: v["avg_us"] for k, v in d["kernels"].items()})
except Exception as e: print("variant $v: no line", e)
PY
done
for v in ${TEST_VARIANTS:-0 1 2}; do
  ( HIPDEC_RECON_VARIANT=$v timeout 150 python -m pytest tests/test_decode_gpu.py tests/test_full_shape_gpu.py -m gpu -q -x > gpurun_out/reconv_tests_$v.log 2>&1; echo "tests variant $v rc=$?"; tail -1 gpurun_out/reconv_tests_$v.log ) &
done
wait
