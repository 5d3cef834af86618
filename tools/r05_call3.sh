#!/bin/bash
# round 5, GPU call 3: the GPU tier (look-ahead, libheif tracks, registry bound), sequence fps with motion / pixel steps, kernel trace of one P track
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q > gpurun_out/c3_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/c3_tests.log
for k in 8 16; do
  echo "== HIPDEC_SEQ_LOOKAHEAD=$k"; HIPDEC_SEQ_LOOKAHEAD=$k timeout 300 python tools/sequence_fps.py 33 16 2>&1 | tail -3
done > gpurun_out/c3_seqfps.txt 2>&1
cat gpurun_out/c3_seqfps.txt
( cd /tmp && export TMPDIR=/tmp
  HIPDEC_SEQ_LOOKAHEAD=16 SEQ_KIND=lowdelay timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3_seqprof -o p -- \
     python $GRAFT_REPO_ROOT/tools/sequence_fps.py 33 1 > $GRAFT_REPO_ROOT/gpurun_out/c3_seqprof.txt 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/c3_seqprof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -20 $f | cut -c1-200 )
