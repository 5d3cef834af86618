#!/bin/bash
# round 5, GPU call 16: the bench line with the library as committed (look-ahead 32), then the GPU tier with a time stamp per test (where do its minutes go?)
mkdir -p gpurun_out
T0=$(date +%s)
python -c "import torch" 2>/dev/null
echo "torch import: $(( $(date +%s) - T0 )) s"
T1=$(date +%s)
timeout 420 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
echo "bench rc=$? in $(( $(date +%s) - T1 )) s"; head -c 200 gpurun_out/final_bench.json; echo
T2=$(date +%s)
timeout ${TIER_LIMIT:-330} python -u -m pytest tests -m gpu -v --timeout 300 -p no:cacheprovider 2>&1 | python -u -c "
import sys, time
t0 = time.time()
for l in sys.stdin:
    sys.stdout.write('%7.1f %s' % (time.time() - t0, l)); sys.stdout.flush()
" > gpurun_out/final_tests_timed.log
echo "tier ended after $(( $(date +%s) - T2 )) s"; tail -2 gpurun_out/final_tests_timed.log
