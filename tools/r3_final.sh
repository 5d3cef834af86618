#!/bin/bash
# round-3 final GPU run within the remaining budget: the whole GPU test tier, the default bench line, one rocprofv3 kernel-stats pass over the main
# workload (the PMC passes of final_run.sh are left out: profiles/pmc_traffic.json and r03z_pmc_parse_b512.txt are of commit 1e6509c, whose parse
# kernel is this one's) -> gpurun_out/final_*
mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -q > gpurun_out/final_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/final_tests.log
timeout 700 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/final_bench.err; head -c 400 gpurun_out/final_bench.json; echo
( cd /tmp && export TMPDIR=/tmp
  HIPDEC_SYNC_UPLOAD=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_prof -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --only-main --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/final_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/final_prof.err
  echo "prof rc=$?"; find $GRAFT_REPO_ROOT/gpurun_out/final_prof -name '*kernel_stats.csv' | head -2 )
