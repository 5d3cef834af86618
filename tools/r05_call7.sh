#!/bin/bash
# round 5, GPU call 7: sequence fps after the early exits in k_motion (+ kernel stats of one P track), sequence GPU tests
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_sequence_gpu.py -m gpu -q --timeout 200 > gpurun_out/c7_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c7_tests.log | cut -c1-200
for k in 16 32; do
  echo "== HIPDEC_SEQ_LOOKAHEAD=$k"; HIPDEC_SEQ_LOOKAHEAD=$k timeout 300 python tools/sequence_fps.py 33 16 2>&1 | tail -3
done > gpurun_out/c7_seqfps.txt 2>&1
cat gpurun_out/c7_seqfps.txt
( cd /tmp && export TMPDIR=/tmp
  SEQ_KIND=lowdelay timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c7_seqprof -o p -- \
     python $GRAFT_REPO_ROOT/tools/sequence_fps.py 33 1 > $GRAFT_REPO_ROOT/gpurun_out/c7_seqprof.txt 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/c7_seqprof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -7 $f | cut -c1-160
  SEQ_KIND=unrestricted timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c7_seqprof_b -o p -- \
     python $GRAFT_REPO_ROOT/tools/sequence_fps.py 33 1 > $GRAFT_REPO_ROOT/gpurun_out/c7_seqprof_b.txt 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/c7_seqprof_b -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -5 $f | cut -c1-160 )
