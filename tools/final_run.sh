#!/bin/bash
# round-end GPU run: the whole GPU test tier, the default bench line, a rocprofv3 kernel-stats pass over the main workload, the parser's PMC
# passes and the HBM traffic passes -> gpurun_out/final_* (copied to profiles/r<NN>z_* by hand)
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null   # (a cold import takes minutes on a fresh box: not inside the tier's time limit)
timeout ${FINAL_TESTS_LIMIT:-500} python -m pytest tests -m gpu -q --timeout 400 > gpurun_out/final_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/final_tests.log
timeout ${FINAL_BENCH_LIMIT:-600} python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
echo "bench rc=$?"; tail -c 300 gpurun_out/final_bench.err; head -c 300 gpurun_out/final_bench.json; echo
( cd /tmp && export TMPDIR=/tmp
  HIPDEC_SYNC_UPLOAD=1 timeout ${FINAL_PROF_LIMIT:-200} rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_prof -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --only-main --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/final_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/final_prof.err
  echo "prof rc=$?"; find $GRAFT_REPO_ROOT/gpurun_out/final_prof -name '*kernel_stats.csv' | head -2 )
bash tools/prof_parse_pmc.sh final --batch 512 > gpurun_out/final_pmc.txt 2>&1; grep k_parse gpurun_out/final_pmc.txt | cut -c1-400
bash tools/prof_hbm_traffic.sh 2>&1 | tail -3
